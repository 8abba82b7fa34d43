# Same-box A/B of a kernel change: builds the library of the last COMMIT into tools/oldlib/ (git-ignored, travels with a gpurun snapshot);
# on the GPU box alternate   LD_LIBRARY_PATH=$PWD/tools/oldlib tools/cbench ...   and   tools/cbench ...   (the working tree's library).
# Boxes differ by +-3 % among themselves; this is how the +0.8 % of round 3's twiddle prefetch was seen (57.7 -> 58.2K, three alternating runs).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/oldsrc "$R/tools/oldlib"; mkdir -p /tmp/oldsrc "$R/tools/oldlib"
git -C "$R" archive HEAD dspfun_amd/csrc include | tar -x -C /tmp/oldsrc
make -s -j8 -C /tmp/oldsrc/dspfun_amd/csrc
cp /tmp/oldsrc/dspfun_amd/csrc/libdspfft_hip.so "$R/tools/oldlib/"
rm -rf /tmp/oldsrc
echo "tools/oldlib/libdspfft_hip.so = $(git -C "$R" rev-parse --short HEAD)"
