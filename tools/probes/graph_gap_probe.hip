// tools/probes/graph_gap_probe.hip -- what does a hipGraph save between DEPENDENT kernels on this device?  (round 6)
// A render is a chain of kernels that each depend on the one before (primary -> trace -> shade -> trace -> ...): 15 launches per C3 render, ~60 at the
// reference's queue size, ~6 us apart in rocprofv3's timeline.  The host is already ahead of the device there (tyr_render queues an iteration before the
// counts of the one before have arrived), so what a graph could save is the part of that gap that is NOT the device's own kernel-to-kernel dependency.
// This probe times a chain of N dependent kernels of ~T us each, launched one by one on a stream (the host far ahead: nothing waits for it) and as one
// captured graph, with hipEvents around the chain: (time - N * T) / N is the gap per kernel either way.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/graph_gap_probe.hip -o /tmp/graph_gap_probe && /tmp/graph_gap_probe
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>

#define CK(x)                                                                      \
	do {                                                                           \
		hipError_t e_ = (x);                                                       \
		if (e_ != hipSuccess) {                                                    \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
			std::exit(1);                                                          \
		}                                                                          \
	} while (0)

__global__ void k_spin(unsigned long long ticks, unsigned int* sink) {
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
	while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
	}
	if (threadIdx.x == 0 && blockIdx.x == 0)
		atomicAdd(sink, 1u);
}

int main(int argc, char** argv) {
	const int n = argc > 1 ? std::atoi(argv[1]) : 200;
	unsigned int* sink = nullptr;
	CK(hipMalloc(&sink, 4));
	CK(hipMemset(sink, 0, 4));
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	for (int blocks : { 1, 512 }) {
		for (unsigned long long us : { 2ull, 50ull, 300ull }) {
			const unsigned long long ticks = us * 100ull;
			auto chain = [&]() {
				for (int i = 0; i < n; ++i)
					hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, s, ticks, sink);
			};
			chain(); // warm
			CK(hipStreamSynchronize(s));
			float msStream = 1e30f, msGraph = 1e30f;
			for (int rep = 0; rep < 5; ++rep) {
				CK(hipEventRecord(a, s));
				chain();
				CK(hipEventRecord(b, s));
				CK(hipEventSynchronize(b));
				float ms;
				CK(hipEventElapsedTime(&ms, a, b));
				msStream = ms < msStream ? ms : msStream;
			}
			hipGraph_t g;
			hipGraphExec_t ge;
			CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
			chain();
			CK(hipStreamEndCapture(s, &g));
			CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
			CK(hipGraphLaunch(ge, s));
			CK(hipStreamSynchronize(s));
			for (int rep = 0; rep < 5; ++rep) {
				CK(hipEventRecord(a, s));
				CK(hipGraphLaunch(ge, s));
				CK(hipEventRecord(b, s));
				CK(hipEventSynchronize(b));
				float ms;
				CK(hipEventElapsedTime(&ms, a, b));
				msGraph = ms < msGraph ? ms : msGraph;
			}
			CK(hipGraphExecDestroy(ge));
			CK(hipGraphDestroy(g));
			std::printf("%3d blocks x %3llu us x %d dependent kernels: stream %.3f ms (gap %.2f us), graph %.3f ms (gap %.2f us)\n", blocks, us, n, msStream, (msStream * 1e3 - n * (double)us) / n, msGraph,
			            (msGraph * 1e3 - n * (double)us) / n);
		}
	}
	return 0;
}
