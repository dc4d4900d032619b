// Shared host/device declarations of the batched ksw_extd2 path (K0/K1/K2 of SURVEY.md 8a).
#pragma once
#include <stdint.h>

#define GD_NEG_INF (-0x40000000)

// which DP kernel owns a task
enum : int32_t {
	GD_KIND_GENERIC = 0, // LDS-resident literal kernel, any geometry that fits the LDS window
	GD_KIND_WAVE64  = 1, // register-resident, one 64-lane wavefront per alignment (<= 64 16-cell blocks in flight)
	GD_KIND_WAVE16  = 2, // register-resident, four alignments per wavefront (<= 16 blocks: short reads)
	GD_KIND_WAVE128 = 3, // register-resident, two 16-cell blocks per lane (<= 128 blocks: ONT bands)
};

// One alignment of a batch.  Built on the host from the lengths, read by every kernel of the batch.
struct KswTask {
	int64_t qoff, toff;   // byte offsets of query / target in the packed nt4 buffers
	int64_t bt_off;       // byte offset of this alignment's backtrace matrix in the arena
	int64_t cig_off;      // offset (in uint32 ops) of this alignment's CIGAR slot
	int32_t qlen, tlen;
	int32_t w;            // band width as passed by the caller (may be < 0)
	int32_t row_bytes;    // backtrace row stride in bytes (generic: n_col_*16 of the reference; wave: 16*lanes)
	int32_t cig_cap;      // capacity of the CIGAR slot in ops
	int32_t exact_score;  // GD_NEG_INF: no exact-match pre-filter
	int32_t kind;         // GD_KIND_*
	int32_t pad;
};

// Scoring constants after the reference's normalisation (SR/ksw2_extd2_sse.c:78-105)
struct KswConst {
	int32_t q, e, q2, e2;          // swapped so that q+e <= q2+e2 (:78)
	int32_t sc_mch, sc_mis, sc_N;  // mat[0], mat[1], (mat[24]==0 ? -e2 : mat[24])   (:85-87)
	int32_t long_thres, long_diff; // :102-105
};

// per-alignment status written by the kernels
enum : int32_t { GD_ST_PENDING = 0, GD_ST_EXACT = 1, GD_ST_DONE = 2, GD_ST_ZDROPPED = 3,
                 GD_ST_TRACED = 4 }; // DP and backtrack both done (the 64-lane kernel walks its own alignment back)

// band of anti-diagonal r (SR/ksw2_extd2_sse.c:138-141); returns st0 > en0 when the band is empty
static inline __host__ __device__ void gd_band(int r, int qlen, int tlen, int w, int &st0, int &en0)
{
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < ((r - w + 1) >> 1)) st = (r - w + 1) >> 1;
	if (en > ((r + w) >> 1)) en = (r + w) >> 1;
	st0 = st, en0 = en;
}

// n_col_ of the reference (SR/ksw2_extd2_sse.c:92-95): number of 16-lane blocks a backtrace row may hold
static inline __host__ __device__ int gd_ncol16(int qlen, int tlen, int w)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int n = qlen < tlen ? qlen : tlen;
	return ((n < w + 1 ? n : w + 1) + 15) / 16 + 1;
}
