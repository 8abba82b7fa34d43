// zcopy.hip -- the memory floor of motion's z pass (BASELINE config 5 as a 3-D volume: 256 frames of 1920 x 1080 floats, 2.1 GB, HBM-resident): the volume
// copied in place in tiles of 256 frames x K floats (one 4K-byte... one K*4-byte segment per frame, a frame apart), 256 threads, lane = 16 bytes, no LDS, no
// arithmetic -- against the frame pitch (floats between frames): does padding the pitch move the segments of a tile onto more memory channels?
//   hipcc -O3 --offload-arch=gfx950 tools/zcopy.hip -o tools/zcopy && tools/zcopy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int K, int T, int D>
__global__ void __launch_bounds__(T) z_copy(const float4 *in, float4 *out, long long pitch4, int ntiles, int remap)
{
	constexpr int NP = K / 4, ROUNDS = D * NP / T;
	int b = blockIdx.x;
	const int per = ntiles >> 3, full = per << 3;
	if (remap && b < full) b = (b & 7) * per + (b >> 3);          // neighbouring tiles on one XCD
	float4 v[ROUNDS];
#pragma unroll
	for (int i = 0; i < ROUNDS; i++) { const int it = threadIdx.x + i * T, z = it / NP, jp = it - z * NP; v[i] = in[(size_t)z * pitch4 + (size_t)b * NP + jp]; }
#pragma unroll
	for (int i = 0; i < ROUNDS; i++) { const int it = threadIdx.x + i * T, z = it / NP, jp = it - z * NP; float4 w = v[i]; w.x += 1.f; out[(size_t)z * pitch4 + (size_t)b * NP + jp] = w; }
}
template <int K, int T>
static void run(float *x, long long frame, long long pad, int remap)
{
	constexpr int D = 256;
	const int ntiles = (int)(frame / K);
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto go = [&]() { hipLaunchKernelGGL((z_copy<K, T, D>), dim3(ntiles), dim3(T), 0, 0, (const float4 *)x, (float4 *)x, (frame + pad) / 4, ntiles, remap); };
	for (int i = 0; i < 2; i++) go();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0, 0));
	for (int i = 0; i < 6; i++) go();
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / 6;
	printf("K = %2d floats (%3d-B segments), %d threads, pitch = frame + %7lld B, %s   %8.1f us  %5.2f TB/s\n", K, K * 4, T, pad * 4, remap ? "tiles grouped per XCD" : "tiles round-robin  ", us,
	       2.0 * D * (double)frame * 4 / us / 1e6);
}
int main()
{
	const long long frame = 1920ll * 1080;
	const long long maxpad = 1 << 20;
	float *x; CK(hipMalloc(&x, (size_t)(frame + maxpad) * 256 * 4)); CK(hipMemset(x, 0, (size_t)(frame + maxpad) * 256 * 4));
	for (int remap = 1; remap >= 0; remap--) {
		for (long long pad : {0ll, 64ll, 256ll, 1024ll, 4096ll, 16384ll + 64, 262144ll + 1024 + 64}) {
			run<16, 256>(x, frame, pad, remap);
		}
	}
	for (long long pad : {0ll, 64ll, 1024ll, 16384ll + 64}) { run<32, 256>(x, frame, pad, 1); run<64, 256>(x, frame, pad, 1); run<64, 1024>(x, frame, pad, 1); }
	return 0;
}
