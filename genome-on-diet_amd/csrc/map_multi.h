// Single-process fan-out of one mini-batch over several GPUs: "each mini-batch from step 0 is split into contiguous
// read ranges, one per GPU (balanced by DP cost, not read count); the index is replicated; results are gathered on the host and
// emitted in input order".  This is the form the reference's one-process CLI can use behind step 1 of worker_pipeline
// (LR/map.c:2132-2137; step 2 stays ordered, LR/kthread.c:97-121): one caller thread per context, each mapping its range straight
// into the caller's n_regs / regs arrays at the range's offset -- no copy, no collective.  (bench.py's one-process-per-GPU form uses
// the same split through the Python mirror, genome-on-diet_amd/shard.py.)
#pragma once

// contiguous ranges of equal DP cost: cost of a read = its DP cells for a full-length candidate, (2 len - 1) * min(band + 1, len);
// bounds[0] = 0 <= bounds[1] <= ... <= bounds[n_parts] = n_reads (the boundary before part r is the first read at which the running
// cost reaches r / n_parts of the total -- shard.read_ranges_by_cost, value for value)
extern "C" int gdiet_hip_read_ranges_by_cost(int n_reads, const int32_t *lens, int n_parts, int32_t band, int32_t *bounds)
{
	if (n_reads < 0 || n_parts < 1 || !bounds || (n_reads > 0 && !lens) || band < 0) return GDIET_E_PARAM;
	std::vector<int64_t> cum((size_t)n_reads + 1, 0);
	for (int i = 0; i < n_reads; ++i) {
		const int64_t l = lens[i] > 0 ? lens[i] : 0;
		const int64_t rows = 2 * l - 1 > 0 ? 2 * l - 1 : 0;
		cum[i + 1] = cum[i] + rows * std::min<int64_t>((int64_t)band + 1, l);
	}
	const int64_t total = cum[n_reads];
	bounds[0] = 0;
	for (int r = 1; r < n_parts; ++r) {
		const double target = (double)(total * r) / (double)n_parts;
		const int at = (int)(std::lower_bound(cum.begin(), cum.end(), target, [](int64_t a, double t) { return (double)a < t; }) - cum.begin());
		bounds[r] = std::max(bounds[r - 1], std::min(at, n_reads));
	}
	bounds[n_parts] = n_reads;
	return GDIET_OK;
}

extern "C" int gdiet_hip_map_batch_multi(int n_ctx, gdiet_ctx *const *ctxs, const gdiet_index *const *idxs, const gdiet_mapopt_t *opt, int n_reads,
                                         const char *const *seqs, const int32_t *lens, int32_t *n_regs, gdiet_reg_t **regs)
{
	if (n_ctx < 1 || !ctxs || !idxs || !ctxs[0]) return GDIET_E_PARAM;
	for (int k = 0; k < n_ctx; ++k)
		if (!ctxs[k] || !idxs[k]) { ctxs[0]->err = "gdiet_hip_map_batch_multi: NULL context or index"; return GDIET_E_PARAM; }
	if (!opt || n_reads < 0 || (n_reads > 0 && (!seqs || !lens || !n_regs || !regs))) { ctxs[0]->err = "NULL argument"; return GDIET_E_PARAM; }
	if (n_reads == 0) return GDIET_OK;
	if (n_ctx == 1) return gdiet_hip_map_batch(ctxs[0], idxs[0], opt, n_reads, seqs, lens, n_regs, regs);
	std::vector<int32_t> bounds((size_t)n_ctx + 1);
	const int32_t band = (opt->flag & GD_F_SR) ? std::max(opt->bw_min, 1) : (int32_t)opt->bw;
	int rc = gdiet_hip_read_ranges_by_cost(n_reads, lens, n_ctx, band, bounds.data());
	if (rc) return rc;
	for (int i = 0; i < n_reads; ++i) n_regs[i] = 0, regs[i] = nullptr;
	std::vector<int> rcs((size_t)n_ctx, GDIET_OK);
	std::vector<std::thread> th;
	for (int k = 0; k < n_ctx; ++k) {
		const int lo = bounds[k], hi = bounds[k + 1];
		ctxs[k]->failed_last = 0;
		if (hi <= lo) continue;
		th.emplace_back([=, &rcs]() { rcs[k] = gdiet_hip_map_batch(ctxs[k], idxs[k], opt, hi - lo, seqs + lo, lens + lo, n_regs + lo, regs + lo); });
	}
	for (auto &t : th) t.join();
	// errors joined: the first failing range decides the return code, its text (with the range) goes to ctxs[0]; nothing half-done is
	// handed back
	for (int k = 0; k < n_ctx; ++k)
		if (rcs[k]) {
			char head[96];
			snprintf(head, sizeof head, "context %d of %d (device %d, reads %d..%d): ", k, n_ctx, ctxs[k]->device, bounds[k], bounds[k + 1] - 1);
			const std::string text = head + ctxs[k]->err;
			gdiet_hip_free_regs(n_reads, n_regs, regs);
			ctxs[0]->err = text;
			return rcs[k];
		}
	if (n_ctx > 1) { // the per-read failures of the other ranges are reported through ctxs[0] as well
		int64_t f = 0;
		for (int k = 0; k < n_ctx; ++k) f += ctxs[k]->failed_last;
		for (int k = 1; k < n_ctx; ++k)
			if (ctxs[k]->failed_last) ctxs[0]->warn = ctxs[k]->warn, ctxs[0]->failed_total += ctxs[k]->failed_last;
		ctxs[0]->failed_last = f;
	}
	return GDIET_OK;
}

extern "C" int gdiet_hip_map_failed_reads(const gdiet_ctx *ctx, int64_t *last_call, int64_t *total, const char **what)
{
	if (!ctx) return GDIET_E_PARAM;
	if (last_call) *last_call = ctx->failed_last;
	if (total) *total = ctx->failed_total;
	if (what) *what = ctx->warn.c_str();
	return GDIET_OK;
}
