// ntbench.hip -- experiment: does a streaming copy of an HBM-resident buffer (398 MB, an 8K frame) run faster with non-temporal
// loads / stores (global_load_dwordx4 ... nt) than with plain ones?  Build: hipcc --offload-arch=gfx950 -O3 tools/ntbench.hip -o tools/ntbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) copy_k(const f4 *src, f4 *dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		f4 v;
		if (MODE & 1) v = __builtin_nontemporal_load(src + i); else v = src[i];
		v.x += 1.f;
		if (MODE & 2) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
	}
}
int main()
{
	const size_t bytes = (size_t)7680 * 4320 * 3 * 4, n = bytes / 16;
	f4 *a, *b; CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes));
	CHK(hipMemset(a, 0, bytes)); CHK(hipMemset(b, 0, bytes));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const char *names[] = {"plain load, plain store", "nt load, plain store", "plain load, nt store", "nt load, nt store"};
	for (int inplace = 0; inplace < 2; inplace++)
		for (int mode = 0; mode < 4; mode++) {
			float best = 1e9;
			for (int rep = 0; rep < 5; rep++) {
				CHK(hipEventRecord(e0));
				f4 *d = inplace ? a : b;
				switch (mode) {
				case 0: hipLaunchKernelGGL(copy_k<0>, dim3(256 * 16), dim3(256), 0, 0, a, d, n); break;
				case 1: hipLaunchKernelGGL(copy_k<1>, dim3(256 * 16), dim3(256), 0, 0, a, d, n); break;
				case 2: hipLaunchKernelGGL(copy_k<2>, dim3(256 * 16), dim3(256), 0, 0, a, d, n); break;
				default: hipLaunchKernelGGL(copy_k<3>, dim3(256 * 16), dim3(256), 0, 0, a, d, n); break;
				}
				CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
				float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
			}
			printf("%s %-26s %7.1f us = %.2f TB/s (read + write)\n", inplace ? "in place    " : "out of place", names[mode], best * 1e3, 2.0 * bytes / best / 1e9);
		}
	return 0;
}
