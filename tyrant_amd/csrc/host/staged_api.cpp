// staged_api.cpp -- test hooks behind the C ABI (kernel-level parity, no product logic): one stage of an iteration at a time
// (tyr_stage_*), and the reference's AoS queue records in and out (tyr_queue_import / _export, tyr_shadow_import / _export: the
// library sorts its physically unordered queues by virtual slot to hand out the reference's order).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "driver_internal.hpp"

using namespace tyr;
using namespace tyr::drv;

extern "C" {

// ---- stage-level API -----------------------------------------------------------------------
int tyr_stage_begin(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	if (!c->haveScene)
		return TYR_ERR_NO_SCENE;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	if ((rc = stage_begin(c)))
		return rc;
	return sync_counters(c);
}
int tyr_stage_primary(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	const uint32_t nNew = planned_new(c), nLive = c->hK->primary_ray_cnt + nNew;
	(void)nLive;
	enqueue_primary(c, make_params(c), nNew);
	HIPCHK(hipGetLastError());
	rc = sync_counters(c);
	collect_timings(c);
	return rc;
}
int tyr_stage_extend(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	enqueue_extend(c, make_params(c), c->hK->n_live, c->hK->n_live); // the host mirror no longer has the survivor count: upper bound
	HIPCHK(hipGetLastError());
	rc = sync_counters(c);
	collect_timings(c);
	return rc ? rc : check_device_error(c);
}
int tyr_stage_shade(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	enqueue_shade(c, make_params(c), c->hK->n_live);
	HIPCHK(hipGetLastError());
	rc = sync_counters(c);
	collect_timings(c);
	return rc ? rc : check_device_error(c);
}
int tyr_stage_connect(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	enqueue_connect(c, make_params(c), c->hK->shadow_ray_cnt);
	HIPCHK(hipGetLastError());
	rc = sync_counters(c);
	collect_timings(c);
	return rc ? rc : check_device_error(c);
}
int tyr_stage_end(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	stage_end(c);
	return TYR_OK;
}
int tyr_sync(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	return sync_counters(c);
}

// ---- AoS import / export (fixtures, parity tests) -------------------------------------------
// The device's queues are physically unordered (hip/kernels.hpp "Queues"); the ABI's queues are the reference's: record i
// is the ray in slot i of the serial order.  Export gathers the records the segments hold and sorts them by virtual slot
// (survivors of the previous iteration first, in the order of the slots they had there -- their rank -- then this
// iteration's primary rays by ticket); import lays the records down in order, slot = position.
int tyr_queue_export(tyr_ctx* c, int which, tyr_ray_queue* host, uint32_t count) {
	if (!c || !host || (which != 0 && which != 1) || count > c->cfg.queue_size)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	const int qi = which == 0 ? c->cur : (c->cur ^ 1);
	const RayQ& q = c->q[qi];
	std::vector<uint32_t> slots;
	for (uint32_t cls = 0; cls < tyr::kClasses; ++cls) { // both classes: where a record lies says nothing about its place in the order
		std::vector<uint32_t> part;
		if ((rc = valid_slots(&c->dK->seg[qi][cls][0], part)))
			return rc;
		for (uint32_t sl : part)
			slots.push_back(cls * c->segCap * tyr::kSegs + sl);
	}
	uint32_t extent = 0;
	for (uint32_t sl : slots)
		extent = std::max(extent, sl + 1);
	std::vector<float4> a, d;
	std::vector<float2> b, h;
	std::vector<uint32_t> f, key;
	if ((rc = gather(q.o_dx, slots, extent, a)) || (rc = gather(q.dyz, slots, extent, b)) || (rc = gather(q.direct_ix, slots, extent, d)) || (rc = gather(q.flags, slots, extent, f)) ||
	    (rc = gather(q.hit, slots, extent, h)) || (rc = gather(q.key, slots, extent, key)))
		return rc;
	std::vector<uint32_t> order(slots.size());
	for (uint32_t i = 0; i < order.size(); ++i)
		order[i] = i;
	auto rankOf = [&](uint32_t i) { return (static_cast<uint64_t>((key[i] & tyr::kKeyIndirect) ? 0u : 1u) << 32) | (key[i] & tyr::kKeyMask); };
	std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return rankOf(x) < rankOf(y); });
	std::memset(host, 0, sizeof(tyr_ray_queue) * count);
	for (uint32_t k = 0; k < count && k < order.size(); ++k) {
		const uint32_t i = order[k];
		tyr_ray_queue& r = host[k];
		r.origin[0] = a[i].x;
		r.origin[1] = a[i].y;
		r.origin[2] = a[i].z;
		r.direction[0] = a[i].w;
		r.direction[1] = b[i].x;
		r.direction[2] = b[i].y;
		r.direct[0] = d[i].x;
		r.direct[1] = d[i].y;
		r.direct[2] = d[i].z;
		std::memcpy(&r.index, &d[i].w, 4);
		r.bounces = static_cast<int32_t>(f[i] & 0xffu);
		r.lastSpecular = static_cast<uint8_t>((f[i] >> 8) & 1u);
		r.distance = h[i].x;
		uint32_t id;
		std::memcpy(&id, &h[i].y, 4);
		r.geometry_type = (id & kHitSphere) ? 0 : 1;
		r.identifier = static_cast<int32_t>(id & ~kHitSphere);
	}
	return TYR_OK;
}

// Test hook: the device's OWN rank tables against the order tyr_queue_export presents.  The export sorts the records by
// their key on the host; the kernels never sort -- k_shade turns a key into the ray's slot with v_lookup() over the scan
// tables of the iteration before (hip/device_common.hpp).  Here the same three-part sum is taken from copies of those
// tables for every record of the queue, and compared with the record's place in the sorted order: a wrong table shows
// here, not one iteration later as wrong random numbers.
int tyr_queue_rank_check(tyr_ctx* c, int which, uint32_t* checked_out, uint32_t* mismatches_out) {
	if (!c || (which != 0 && which != 1) || !checked_out || !mismatches_out)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	const int qi = which == 0 ? c->cur : (c->cur ^ 1);
	// the tables the keys of this queue point into: written by the scan of the iteration that made its survivors
	const int t = which == 1 ? static_cast<int>(c->iter & 1u) : static_cast<int>((c->iter & 1u) ^ 1u);
	const size_t N = c->cfg.queue_size, entries = (N + 63) / 64 + kBlock, blocks = (N + 16383) / 16384 + 1;
	std::vector<unsigned long long> word(entries);
	std::vector<uint32_t> pre(entries), blk(blocks);
	HIPCHK(hipMemcpy(word.data(), c->vWord[t], entries * 8, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(pre.data(), c->vPre[t], entries * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(blk.data(), c->vBlk[t], blocks * 4, hipMemcpyDeviceToHost));
	std::vector<uint32_t> keys;
	for (uint32_t cls = 0; cls < tyr::kClasses; ++cls) {
		std::vector<uint32_t> part, k;
		if ((rc = valid_slots(&c->dK->seg[qi][cls][0], part)))
			return rc;
		for (uint32_t& sl : part)
			sl += cls * c->segCap * tyr::kSegs;
		uint32_t extent = 0;
		for (uint32_t sl : part)
			extent = std::max(extent, sl + 1);
		if ((rc = gather(c->q[qi].key, part, extent, k)))
			return rc;
		keys.insert(keys.end(), k.begin(), k.end());
	}
	auto sortKey = [&](uint32_t key) { return (static_cast<uint64_t>((key & tyr::kKeyIndirect) ? 0u : 1u) << 32) | (key & tyr::kKeyMask); };
	std::sort(keys.begin(), keys.end(), [&](uint32_t x, uint32_t y) { return sortKey(x) < sortKey(y); });
	uint32_t bad = 0;
	for (uint32_t i = 0; i < keys.size(); ++i) {
		const uint32_t v = keys[i] & tyr::kKeyMask;
		uint32_t slot = v; // a fresh primary ray carries its slot itself
		if (keys[i] & tyr::kKeyIndirect) {
			const uint32_t e = v >> 6;
			if (e >= entries || (e >> 8) >= blocks) {
				++bad;
				continue;
			}
			slot = blk[e >> 8] + pre[e] + static_cast<uint32_t>(__builtin_popcountll(word[e] & ((1ull << (v & 63u)) - 1ull)));
			if (!((word[e] >> (v & 63u)) & 1ull))
				++bad; // the record's own survive bit must be set
		}
		if (slot != i)
			++bad;
	}
	*checked_out = static_cast<uint32_t>(keys.size());
	*mismatches_out = bad;
	return TYR_OK;
}

int tyr_queue_import(tyr_ctx* c, const tyr_ray_queue* host, uint32_t n) {
	if (!c || (!host && n) || n > c->cfg.queue_size)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	c->lastShadeFolded = false; // imported rays carry no sphere record: the pre-pass kernels do them
	const RayQ& q = c->q[c->cur];
	std::vector<float4> a(n), d(n);
	std::vector<float2> b(n), h(n);
	std::vector<uint32_t> f(n), key(n);
	for (uint32_t i = 0; i < n; ++i) {
		const tyr_ray_queue& r = host[i];
		a[i] = make_float4(r.origin[0], r.origin[1], r.origin[2], r.direction[0]);
		b[i] = make_float2(r.direction[1], r.direction[2]);
		float ix;
		std::memcpy(&ix, &r.index, 4);
		d[i] = make_float4(r.direct[0], r.direct[1], r.direct[2], ix);
		f[i] = (static_cast<uint32_t>(r.bounces) & 0xffu) | ((r.lastSpecular ? 1u : 0u) << 8);
		const uint32_t id = (r.geometry_type == 0 ? kHitSphere : 0u) | static_cast<uint32_t>(r.identifier);
		float idf;
		std::memcpy(&idf, &id, 4);
		h[i] = make_float2(r.distance, idf);
		key[i] = i; // slot = position; no kKeySphereDone: extend's pre-pass computes the sphere half as for any survivor
	}
	if (n) {
		HIPCHK(hipMemcpy(q.o_dx, a.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.dyz, b.data(), n * sizeof(float2), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.direct_ix, d.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.flags, f.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.hit, h.data(), n * sizeof(float2), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(q.key, key.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	c->hK->primary_ray_cnt = n;
	// all of them in class 0 (the traversal's own root test sorts out those that miss the tree)
	dense_counts(n, &c->hK->seg[c->cur][0][0]);
	std::memset(&c->hK->seg[c->cur][1][0], 0, sizeof c->hK->seg[0][0]);
	for (uint32_t w = 0; w < tyr::kSegs; ++w) {
		c->hK->segSurv[0][w] = c->hK->seg[c->cur][0][w * tyr::kSegStride];
		c->hK->segSurv[1][w] = 0;
	}
	return push_counters(c);
}

int tyr_shadow_export(tyr_ctx* c, tyr_shadow_queue* host, uint32_t count) {
	if (!c || !host || count > c->cfg.queue_size)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	std::vector<uint32_t> slots;
	if ((rc = valid_slots(&(c->dKc + c->shadowSet)->seg[0], slots))) // the set of the iteration that was shaded last
		return rc;
	uint32_t extent = 0;
	for (uint32_t s : slots)
		extent = std::max(extent, s + 1);
	std::vector<float4> a, b, col;
	std::vector<uint32_t> key;
	const ShadowQ& sq = c->shadow[c->shadowSet];
	if ((rc = gather(sq.o_dx, slots, extent, a)) || (rc = gather(sq.dyz_cd_ix, slots, extent, b)) || (rc = gather(sq.color, slots, extent, col)) || (rc = gather(sq.key, slots, extent, key)))
		return rc;
	std::vector<uint32_t> order(slots.size());
	for (uint32_t i = 0; i < order.size(); ++i)
		order[i] = i;
	std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y]; }); // the emitting rays' slots: the serial order
	std::memset(host, 0, sizeof(tyr_shadow_queue) * count);
	for (uint32_t k = 0; k < count && k < order.size(); ++k) {
		const uint32_t i = order[k];
		tyr_shadow_queue& s = host[k];
		s.origin[0] = a[i].x;
		s.origin[1] = a[i].y;
		s.origin[2] = a[i].z;
		s.direction[0] = a[i].w;
		s.direction[1] = b[i].x;
		s.direction[2] = b[i].y;
		s.closestDistance = b[i].z;
		std::memcpy(&s.buffer_index, &b[i].w, 4);
		s.color[0] = col[i].x;
		s.color[1] = col[i].y;
		s.color[2] = col[i].z;
	}
	return TYR_OK;
}

int tyr_shadow_import(tyr_ctx* c, const tyr_shadow_queue* host, uint32_t n) {
	if (!c || (!host && n) || n > c->cfg.queue_size)
		return TYR_ERR_INVALID;
	int rc = use_device(c);
	if (rc)
		return rc;
	if ((rc = sync_counters(c)))
		return rc;
	std::vector<float4> a(n), b(n), col(n);
	std::vector<uint32_t> key(n);
	for (uint32_t i = 0; i < n; ++i) {
		const tyr_shadow_queue& s = host[i];
		float ix;
		std::memcpy(&ix, &s.buffer_index, 4);
		a[i] = make_float4(s.origin[0], s.origin[1], s.origin[2], s.direction[0]);
		b[i] = make_float4(s.direction[1], s.direction[2], s.closestDistance, ix);
		col[i] = make_float4(s.color[0], s.color[1], s.color[2], 0.0f);
		key[i] = i;
	}
	if (n) {
		const ShadowQ& sq = c->shadow[c->iter & 1u];
		HIPCHK(hipMemcpy(sq.o_dx, a.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(sq.dyz_cd_ix, b.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(sq.color, col.data(), n * sizeof(float4), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(sq.key, key.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	// what shade leaves behind (kernel.cu:416-417): the counts connect reads, in this iteration's set
	c->hK->shadow_ray_cnt = n;
	c->shadowSet = c->iter & 1u;
	c->lastShadeFolded = false; // (imported shadow rays carry no sphere verdict)
	ConnectCounters* kc = c->dKc + (c->iter & 1u);
	uint32_t cnt[tyr::kSegs * tyr::kSegStride];
	dense_counts(n, cnt);
	HIPCHK(hipMemcpy(&kc->shadow_cnt, &n, sizeof(uint32_t), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(&kc->seg[0], cnt, sizeof cnt, hipMemcpyHostToDevice));
	return push_counters(c);
}

} // extern "C"
