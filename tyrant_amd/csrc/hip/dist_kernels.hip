// dist_kernels.hip -- the two copies around the multi-GPU exchange (host/dist.cpp): pack the rows a rank owns out of
// its full-frame accumulation buffer into a contiguous slab, and scatter a slab back into the rows of the complete
// frame.  Pure streaming copies (16 B per lane, coalesced both ways): HBM-bound, 2 x 16 B per pixel moved.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace tyr {

// slab[yl * W + x] = frame[(yl * nranks + rank) * W + x]
__global__ void __launch_bounds__(kBlock) k_pack_rows(const float4* __restrict__ frame, float4* __restrict__ slab, uint32_t W, uint32_t localRows, uint32_t rank, uint32_t nranks) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= W * localRows)
		return;
	const uint32_t yl = i / W, x = i - yl * W;
	slab[i] = frame[(size_t)(yl * nranks + rank) * W + x];
}

// frame[(yl * nranks + r) * W + x] = slabs[r][yl * W + x] for every rank r; `own` replaces the slab of rank `ownRank`
// (the root's own rows never travel)
__global__ void __launch_bounds__(kBlock) k_scatter_rows(const float4* __restrict__ slabs, const float4* __restrict__ own, uint32_t ownRank, float4* __restrict__ frame, uint32_t W, uint32_t localRows,
	uint32_t nranks) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t slabPixels = W * localRows;
	if (i >= slabPixels * nranks)
		return;
	const uint32_t r = i / slabPixels, j = i - r * slabPixels;
	const uint32_t yl = j / W, x = j - yl * W;
	const float4 v = (r == ownRank) ? own[j] : slabs[(size_t)r * slabPixels + j];
	frame[(size_t)(yl * nranks + r) * W + x] = v;
}

void launch_pack_rows(const float4* frame, float4* slab, uint32_t W, uint32_t localRows, uint32_t rank, uint32_t nranks, hipStream_t stream) {
	const uint32_t n = W * localRows;
	hipLaunchKernelGGL(k_pack_rows, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, frame, slab, W, localRows, rank, nranks);
}
void launch_scatter_rows(const float4* slabs, const float4* own, uint32_t ownRank, float4* frame, uint32_t W, uint32_t localRows, uint32_t nranks, hipStream_t stream) {
	const uint32_t n = W * localRows * nranks;
	hipLaunchKernelGGL(k_scatter_rows, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, slabs, own, ownRank, frame, W, localRows, nranks);
}

} // namespace tyr
