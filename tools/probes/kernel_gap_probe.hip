// tools/probes/kernel_gap_probe.hip -- what is the ~6 us between two kernels of a render made of?  (round 6)
// graph_gap_probe.hip: two dependent EMPTY kernels are 1.1-1.5 us apart on this device, graph or stream.  rocprofv3's timeline of a C3 render shows
// ~6 us between k_shade and k_trace_flat.  Candidate: the end of a kernel writes the L2s' dirty lines back (eight XCDs, 4 MB each), and shade has
// just written ~1 GB of queue records.  This probe: a kernel that stores `mb` megabytes (plain or non-temporal stores), then a kernel that reads one
// word; device timestamps (s_memrealtime, 100 MHz) of the first kernel's last wave and of the second kernel's first wave give the gap.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/kernel_gap_probe.hip -o /tmp/kernel_gap_probe && /tmp/kernel_gap_probe
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>

#define CK(x)                                                            \
	do {                                                                 \
		hipError_t e_ = (x);                                             \
		if (e_ != hipSuccess) {                                          \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
			std::exit(1);                                                \
		}                                                                \
	} while (0)

template <bool NT>
__global__ void k_write(float4* buf, size_t n16, unsigned long long* stamps) {
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	if (threadIdx.x == 0)
		atomicMin(&stamps[0], t0);
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
		if (NT) {
			typedef float v4f __attribute__((ext_vector_type(4)));
			const v4f w = { v.x, v.y, v.z, v.w };
			__builtin_nontemporal_store(w, reinterpret_cast<v4f*>(&buf[i]));
		}
		else
			buf[i] = v;
	}
	__builtin_amdgcn_s_waitcnt(0);
	if (threadIdx.x == 0)
		atomicMax(&stamps[1], __builtin_amdgcn_s_memrealtime());
}
__global__ void k_next(const float4* buf, unsigned long long* stamps, float* sink) {
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	if (threadIdx.x == 0)
		atomicMin(&stamps[2], t0);
	if (threadIdx.x == 0 && blockIdx.x == 0)
		*sink = buf[12345].x;
}

int main() {
	const size_t maxBytes = size_t(2) << 30;
	float4* buf = nullptr;
	CK(hipMalloc(&buf, maxBytes));
	unsigned long long* stamps = nullptr;
	CK(hipMalloc(&stamps, 64));
	float* sink = nullptr;
	CK(hipMalloc(&sink, 4));
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	for (int nt = 0; nt < 2; ++nt)
		for (size_t mb : { size_t(0), size_t(1), size_t(8), size_t(32), size_t(128), size_t(1024) }) {
			const size_t n16 = mb * (size_t(1) << 20) / 16;
			double best = 1e30, bestK = 0;
			for (int rep = 0; rep < 6; ++rep) {
				const unsigned long long init[4] = { ~0ull, 0ull, ~0ull, 0ull };
				CK(hipMemcpyAsync(stamps, init, 32, hipMemcpyHostToDevice, s));
				if (nt)
					hipLaunchKernelGGL(k_write<true>, dim3(2048), dim3(256), 0, s, buf, n16, stamps);
				else
					hipLaunchKernelGGL(k_write<false>, dim3(2048), dim3(256), 0, s, buf, n16, stamps);
				hipLaunchKernelGGL(k_next, dim3(512), dim3(256), 0, s, buf, stamps, sink);
				unsigned long long h[4];
				CK(hipMemcpyAsync(h, stamps, 32, hipMemcpyDeviceToHost, s));
				CK(hipStreamSynchronize(s));
				const double gap = ((double)h[2] - (double)h[1]) * 0.01, kern = ((double)h[1] - (double)h[0]) * 0.01;
				if (rep > 0 && gap < best)
					best = gap, bestK = kern;
			}
			std::printf("%s stores, %5zu MB written (kernel %.1f us): %.2f us from its last wave's end to the next kernel's first wave\n", nt ? "non-temporal" : "plain       ", mb, bestK, best);
		}
	return 0;
}
