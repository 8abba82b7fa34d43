# same-box A/B of the 8K passes: the working tree's library against tools/oldlib (tools/ab_oldlib.sh)
for i in 1 2 3; do
  echo "new: $(python3 tools/bench_8k_quick.py 2>/dev/null | grep -v amdgpu | tail -2 | tr '\n' ' ')"
  echo "old: $(DSPFFT_LIB_PATH=$PWD/tools/oldlib/libdspfft_hip.so python3 tools/bench_8k_quick.py 2>/dev/null | grep -v amdgpu | tail -2 | tr '\n' ' ')"
done
