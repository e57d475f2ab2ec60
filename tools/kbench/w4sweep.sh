K=tools/kbench/conv_bench
for args in "32 64 64 0 1" "32 192 64 0 0" "32 128 64 0 0" "32 64 64 0 2" "32 32 32 0 1" "32 96 32 0 0" "32 64 32 0 0" "16 64 64 0 1" "16 128 64 0 0" "16 32 64 0 2" "32 64 64 2 0"; do
  for w in 0 1; do echo -n "W4=$w  "; RGFM_HX2P_W4=$w timeout -k 10 60 $K $args 512 hx2p | tr '\n' ' ' | sed 's/check vs f32 kernel: //'; echo; done
done
