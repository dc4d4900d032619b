// CPU test driver (test infrastructure): runs the product's host/device-shared stage code (map_stages.h) on the HOST,
// with the CPU oracle's ksw_extd2 standing in for the HIP DP kernel, through the product's host post-processing
// (map_host.h) and prints SAM records.  tests/test_map_host.py diffs that against the reference binary
// (oracle/_ref/gdiet_lr_avx) and against the committed golden SAM.  This is how the seeding/voting/geometry/
// post-processing code is validated in a container without a GPU; the GPU path replaces only the executors.
//
//   map_host_main [-x map-hifi|map-ont] [-k K] [-w W] [-Z pat] [-W n] [-i f] [-r bw] [-s min_dp] [-N n] [--vt_dis=..] ... ref.fa reads.fq
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#define __host__
#define __device__
#include "map_index.h"
#include "map_host.h"
#include "map_stages.h"
#include "gdo_ksw2.h"

static bool read_fastx(const char *fn, std::vector<std::string> &names, std::vector<std::string> &seqs, std::vector<std::string> *quals)
{
	FILE *f = fopen(fn, "r");
	if (!f) return false;
	char *line = 0;
	size_t cap = 0;
	ssize_t l;
	int state = 0; // 0: expect header; 1: fasta seq; 2: fastq seq; 3: fastq '+'; 4: fastq qual
	while ((l = getline(&line, &cap, f)) >= 0) {
		while (l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if (state == 0 || state == 1) {
			if (line[0] == '>' || line[0] == '@') {
				std::string nm(line + 1);
				size_t sp = nm.find_first_of(" \t");
				if (sp != std::string::npos) nm.resize(sp);
				names.push_back(nm), seqs.emplace_back();
				if (quals) quals->emplace_back();
				state = line[0] == '>' ? 1 : 2;
			} else if (state == 1) seqs.back() += line;
		} else if (state == 2) {
			if (line[0] == '+') state = 4;
			else seqs.back() += line;
		} else if (state == 4) {
			if (quals) quals->back() += line;
			if (!quals || quals->back().size() >= seqs.back().size()) state = 0;
		}
	}
	free(line);
	fclose(f);
	return true;
}

int main(int argc, char **argv)
{
	GdMapOpt O;
	// mm_mapopt_init defaults (LR/options.c:13-62) + the GDiet-forced values (LR/main.c:169-185)
	O.a = 2, O.b = 4, O.q = 4, O.e = 2, O.q2 = 24, O.e2 = 1, O.bw = 1000, O.best_n = 5, O.k = 15, O.w = 10;
	int min_mid_occ = 10, max_mid_occ = 1000000;
	float mid_occ_frac = 2e-4f;
	const char *Z = "11";
	int W = 2;
	O.flag = GD_F_NO_PRINT_2ND * 0;
	std::vector<const char *> pos;
	const char *mmi_in = nullptr, *mmi_out = nullptr;
	bool trace = false, out_sam = false, stats = false;
	int n_threads = 1;
	bool preset_seen = false, sr_variant = false;
	auto preset = [&](const char *p) -> bool {
		if (!strcmp(p, "map-hifi")) { O.k = 19, O.w = 19, O.a = 1, O.b = 4, O.q = 6, O.q2 = 26, O.e = 2, O.e2 = 1, O.occ_dist = 500, min_mid_occ = 50, max_mid_occ = 500; }
		else if (!strcmp(p, "map-ont")) {}
		else if (!strcmp(p, "sr") || !strcmp(p, "short")) { // SR/options.c:130-148
			sr_variant = true;
			O.k = 21, O.w = 11, O.flag |= GD_F_SR | GD_F_FRAG_MODE | GD_F_NO_PRINT_2ND;
			O.a = 2, O.b = 8, O.q = 12, O.e = 2, O.q2 = 24, O.e2 = 1, O.max_frag_len = 800, O.min_dp_max = 40, O.best_n = 20, O.mid_occ = 1000;
		} else return false;
		return true;
	};
	for (int pass = 0; pass < 2; ++pass) { // pass 0: -x only (as the reference does), pass 1: everything else
		for (int i = 1; i < argc; ++i) {
			std::string a = argv[i];
			auto val = [&](const char *name) -> const char * { // --name=value or --name value
				std::string p = std::string("--") + name;
				if (a.compare(0, p.size() + 1, p + "=") == 0) return argv[i] + p.size() + 1;
				if (a == p && i + 1 < argc) return argv[++i];
				return nullptr;
			};
			const char *v;
			if (a == "-x") {
				const char *p = argv[++i];
				if (pass == 0) {
					preset_seen = true;
					if (!preset(p)) { fprintf(stderr, "unsupported preset %s\n", p); return 2; }
				}
			} else if (a[0] == '-' && a.size() > 2 && a[1] == 'a' && a[2] == 'x') { // -ax <preset> = -a -x <preset>
				const char *p = argv[++i];
				out_sam = true;
				if (pass == 0) {
					preset_seen = true;
					if (!preset(p)) { fprintf(stderr, "unsupported preset %s\n", p); return 2; }
				}
			} else if (a == "-t") { v = argv[++i]; if (pass) n_threads = atoi(v) > 0 ? atoi(v) : 1; }
			else if (a == "-o") ++i;
			else if (a == "-a") { if (pass) out_sam = true; }
			else if (a == "-c") { if (pass) O.flag |= GD_F_OUT_CG; }
			else if (a == "--paf-no-hit") { if (pass) O.flag |= GD_F_PAF_NO_HIT; }
			else if (a == "-k") { v = argv[++i]; if (pass) O.k = atoi(v); }
			else if (a == "-w") { v = argv[++i]; if (pass) O.w = atoi(v); }
			else if (a == "-Z") { v = argv[++i]; if (pass) Z = v; }
			else if (a == "-W") { v = argv[++i]; if (pass) W = atoi(v); }
			else if (a == "-i") { v = argv[++i]; if (pass) { O.max_seeds = (float)strtod(v, 0); if (O.max_seeds < 0) O.max_seeds = 0.1f; } }
			else if (a == "-r") {
				v = argv[++i];
				if (pass && !sr_variant) O.bw = (uint32_t)strtoul(v, 0, 10);
				else if (pass) { // SR/main.c:419-432
					char *e;
					const double x = strtod(v, &e);
					if (x < 1.0) {
						O.bw_frac = (float)x;
						if (*e == ',') { O.bw_min = (int)strtol(e + 1, &e, 10); if (*e == ',') O.bw_max = (int)strtol(e + 1, &e, 10); }
					} else O.bw = (uint32_t)(int)(x + .499);
				}
			}
			else if (a == "-n") { // SR/main.c: -n FLOAT[,FLOAT]
				v = argv[++i];
				if (pass) { char *e; O.min_cnt = strtof(v, &e); if (*e == ',') O.rec_threshold_frac = strtof(e + 1, &e); }
			}
			else if ((v = val("AF_max_loc"))) { if (pass) O.af_max_loc = (int)atof(v); }
			else if (a == "--stats") { if (pass) stats = true; }          // count the events the repeat-rich fixtures exist for (oracle/make_golden.py prints them)
			else if (a == "--print-seeds") { if (pass) trace = true; }   // the reference's stage trace (RS/SD/VT/AVT/BE/AL_SCORE lines, LR/map.c:1328-1338,1447-1459,1592-1602,1670-1675,1808-1810)
			else if ((v = val("mmi"))) { if (pass) mmi_in = v; }           // use an .mmi written by the reference instead of building the index
			else if ((v = val("dump-mmi"))) { if (pass) mmi_out = v; }    // write the index as an .mmi and exit
			else if (a == "-f") { // LR/main.c:440-448: a fraction selects mid_occ from the index, a number >= 1 is mid_occ itself
				v = argv[++i];
				if (pass) { const double x = strtod(v, 0); if (x < 1.0) mid_occ_frac = (float)x, O.mid_occ = 0; else O.mid_occ = (int)(x + .499); }
			}
			else if (a == "-s") { v = argv[++i]; if (pass) O.min_dp_max = atoi(v); }
			else if (a == "-N") { v = argv[++i]; if (pass) O.best_n = atoi(v); }
			else if (a.compare(0, 2, "-F") == 0) { if (a.size() == 2) ++i; }
			else if ((v = val("vt_dis"))) { if (pass) O.vt_dis = (uint32_t)strtoul(v, 0, 10); }
			else if ((v = val("vt_nb_loc"))) { if (pass) O.vt_nb_loc = (uint32_t)strtoul(v, 0, 10); }
			else if ((v = val("vt_cov"))) { if (pass) O.vt_cov = strtof(v, 0); }
			else if ((v = val("vt_f"))) { if (pass) O.vt_f = strtof(v, 0); }
			else if ((v = val("vt_df1"))) { if (pass) O.vt_df1 = strtof(v, 0); }
			else if ((v = val("vt_df2"))) { if (pass) O.vt_df2 = strtof(v, 0); }
			else if ((v = val("max_min_gap"))) { if (pass) O.max_min_gap = (uint32_t)strtoul(v, 0, 10); }
			else if ((v = val("max_max_gap"))) { if (pass) O.max_max_gap = (uint32_t)strtoul(v, 0, 10); }
			else if ((v = val("sort")) || (v = val("frag"))) {}
			else if ((v = val("secondary"))) { if (pass) { if (!strcmp(v, "yes")) O.flag &= ~(int64_t)GD_F_NO_PRINT_2ND; else O.flag |= GD_F_NO_PRINT_2ND; } }
			else if (a[0] == '-') { fprintf(stderr, "unsupported option %s\n", a.c_str()); return 2; }
			else if (pass) pos.push_back(argv[i]);
		}
		if (pass == 0) { // GDiet-forced values after the preset (LR/main.c:169-185)
			O.max_seeds = 0.1f, O.vt_dis = 100, O.vt_nb_loc = 3, O.vt_cov = 0.03f, O.vt_df1 = 0.01f, O.vt_df2 = 0.01f, O.vt_f = 0.05f;
			O.max_max_gap = 50000, O.max_min_gap = 4000;
			if (!sr_variant) O.min_dp_max = 40;
			else O.min_cnt = 1.0f, O.rec_threshold_frac = 0.0f, O.af_max_loc = 20, O.bw = 0, O.bw_min = 500, O.bw_max = 1500, O.bw_frac = 0.05f; // SR/main.c:163-172, SR/options.c:24
		}
	}
	(void)preset_seen;
	if (pos.size() < 2) { fprintf(stderr, "usage: map_host_main [options] ref.fa reads.fq\n"); return 2; }
	if (!gd_pattern_init(O.pat, Z, W)) { fprintf(stderr, "bad pattern\n"); return 2; }
	std::vector<std::string> rn, rs, qn, qs, qq;
	if (!read_fastx(pos[0], rn, rs, nullptr) || !read_fastx(pos[1], qn, qs, &qq)) { fprintf(stderr, "cannot read input\n"); return 2; }
	GdIndex I;
	std::vector<GdSeqSpan> spans(rs.size());
	for (size_t i = 0; i < rs.size(); ++i) spans[i].p = rs[i].data(), spans[i].n = rs[i].size();
	if (mmi_in) {
		std::string err;
		if (!gd_index_read_mmi(I, mmi_in, O.pat, err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
	} else gd_index_build(I, rn, spans, O.k, O.w, O.pat, 8, true);
	if (mmi_out) {
		std::string err;
		if (!gd_index_write_mmi(I, mmi_out, 14, err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
		return 0;
	}
	// mm_mapopt_update (LR/options.c:64-76)
	if (O.mid_occ <= 0) {
		O.mid_occ = gd_index_cal_max_occ(I, mid_occ_frac);
		if (O.mid_occ < min_mid_occ) O.mid_occ = min_mid_occ;
		if (max_mid_occ > min_mid_occ && O.mid_occ > max_mid_occ) O.mid_occ = max_mid_occ;
	}
	fprintf(stderr, "[map_host_main] keys=%llu mid_occ=%d k=%d w=%d\n", (unsigned long long)I.n_keys, O.mid_occ, O.k, O.w);
	const GdIdxView V = I.view();
	const GdRefView R = I.ref();
	const int g = O.a, bb = O.b < 0 ? O.b : -O.b;
	int8_t mat[25];
	for (int i = 0; i < 25; ++i) mat[i] = (i / 5 == 4 || i % 5 == 4) ? 0 : (i / 5 == i % 5 ? (int8_t)g : (int8_t)bb);
	// reads are independent: a few worker threads, results printed in input order (the oracle's scalar DP is the slow part)
	std::vector<std::string> sam_of(qs.size()), trace_of(qs.size());
	// --stats: reads on which each high-occurrence branch of the seeding stage fires
	std::atomic<long> st_mzflt{0}, st_mzflt_dropped{0}, st_high{0}, st_over_max{0}, st_heap_replace{0}, st_big_strand{0}, st_max_hits{0}, st_flt_rescued{0}, st_degenerate{0};
	auto map_one = [&](size_t ri) {
		std::string out;
		char *tbuf = nullptr;
		size_t tlen_ = 0;
		FILE *terr = trace ? open_memstream(&tbuf, &tlen_) : nullptr;
		const std::string &seq = qs[ri];
		const int len = (int)seq.size();
		std::vector<GdReg> regs;
		if (len > 0) {
			std::vector<uint8_t> enc(len), rev(len);
			for (int j = 0; j < len; ++j) enc[j] = gd_nt4((unsigned char)seq[j]);
			for (int j = 0; j < len; ++j) rev[len - 1 - j] = enc[j] ^ 3; // N (4) becomes 7 exactly as qs_rev does (LR/map.c:1634,:1641)
			const unsigned maxm = (unsigned)len + 64;
			std::vector<GdMini> mv(maxm);
			std::vector<uint32_t> shift_n(O.pat.W);
			unsigned tot = gd_sketch2(enc.data(), len, O.w, O.k, O.pat, O.max_seeds, mv.data(), maxm, shift_n.data());
			(void)tot;
			const int shift = (int)gd_get_shift(V, mv.data(), shift_n.data(), O.pat.W);
			if (trace) fprintf(terr, "QR\t%s\nFinal shift: %d\n", qn[ri].c_str(), shift);
			unsigned n_mv = 0;
			const uint32_t cap = (O.flag & GD_F_FRAG_MODE) ? (O.max_frag_len == 0 ? 800u : (uint32_t)O.max_frag_len) : UINT32_MAX;
			const unsigned tel = gd_sketch3(enc.data(), (unsigned)len, O.w, O.k, O.pat, shift, cap, mv.data(), maxm, &n_mv);
			std::vector<uint64_t> scratch(2 * (size_t)n_mv + 2);
			const unsigned n_mv0 = n_mv;
			if (O.q_occ_frac > 0.0f) n_mv = gd_mz_flt(mv.data(), n_mv, O.mid_occ, O.q_occ_frac, scratch.data());
			if (stats) {
				if (n_mv < n_mv0) ++st_mzflt, st_mzflt_dropped += n_mv0 - n_mv;
				// the seeds present in the index, in sketch order: streaks of n > mid_occ (LR/seed.c:66-106)
				std::vector<std::pair<uint32_t, uint32_t>> pr; // (n, q_pos >> 1)
				for (unsigned i = 0; i < n_mv; ++i) {
					uint64_t st_;
					const uint32_t n = gd_idx_get(V, mv[i].x >> 8, &st_);
					if (n) pr.emplace_back(n, (uint32_t)mv[i].y >> 1);
				}
				bool high = false, over = false, repl = false, resc = false;
				int last0 = -1;
				for (int i = 0; i <= (int)pr.size(); ++i)
					if (i == (int)pr.size() || (int32_t)pr[i].first <= O.mid_occ) {
						if (i - last0 > 1) {
							high = true;
							const int ps = last0 < 0 ? 0 : (int)pr[last0].second, pe = i == (int)pr.size() ? len : (int)pr[i].second;
							int mh = (int)((double)(pe - ps) / O.occ_dist + .499);
							if (mh > 128) mh = 128;
							if (mh > 0) resc = true;
							if (mh > 0 && i - last0 - 1 > mh) repl = true;
							for (int j = last0 + 1; j < i; ++j) over |= (int32_t)pr[j].first > O.max_max_occ;
						}
						last0 = i;
					}
				st_high += high, st_over_max += over, st_heap_replace += repl, st_flt_rescued += resc;
			}
			std::vector<GdSeed> seeds(n_mv + 1);
			int64_t n_a = 0;
			const int n_m = gd_collect_matches2(V, mv.data(), n_mv, len, O.mid_occ, O.max_max_occ, O.occ_dist, seeds.data(), &n_a);
			std::vector<GdLoc> af(n_a + 1), ar(n_a + 1), tmp(n_a + 1);
			unsigned nf = 0, nr = 0;
			gd_seed_hits(V, seeds.data(), n_m, O.flag, tel, af.data(), ar.data(), &nf, &nr);
			if (stats) {
				if (nf > 4096 || nr > 4096) ++st_big_strand;
				long cur = st_max_hits.load(), m = nf > nr ? nf : nr;
				while (m > cur && !st_max_hits.compare_exchange_weak(cur, m)) {}
			}
			GdLoc *sf = gd_sort_locs(af.data(), tmp.data(), nf);
			std::vector<GdLoc> sfv(sf, sf + nf);
			GdLoc *sr = gd_sort_locs(ar.data(), tmp.data(), nr);
			std::vector<GdLoc> srv(sr, sr + nr);
			if (trace) {
				fprintf(terr, "RS n_a_for: %u, n_a_rev: %u\n", nf, nr);
				for (unsigned i = 0; i < nf; ++i) fprintf(terr, "SD\t%s\t%d\t+\t%u\n", R.seq[sfv[i].target >> 32].name.c_str(), (int32_t)sfv[i].target + 1 - (int32_t)tel, sfv[i].query);
				for (unsigned i = 0; i < nr; ++i) fprintf(terr, "SD\t%s\t%d\t-\t%u\n", R.seq[srv[i].target >> 32].name.c_str(), (uint32_t)srv[i].target + 1, srv[i].query);
			}
			GdLrVoteOpt VO = {O.vt_dis, O.vt_nb_loc, O.bw, O.vt_cov, O.vt_f, O.vt_df1, O.vt_df2, O.k};
			GdSrVoteOpt SO = {O.min_cnt, O.rec_threshold_frac, O.bw_frac, O.bw_min, O.bw_max, O.af_max_loc, cap, (O.flag & GD_F_FRAG_MODE) != 0};
			GdVt vts[GDM_MAX_VT];
			unsigned nc = sr_variant ? gd_sr_candidates(sfv.data(), nf, srv.data(), nr, (uint32_t)len, tel, n_mv, SO, vts)
			                         : gd_lr_candidates(sfv.data(), nf, srv.data(), nr, (uint32_t)len, (int32_t)tel, VO, vts);
			const int dp_bw = sr_variant ? (int)gd_sr_bw(len, SO) : (int)O.bw;
			if (trace && nc > 0) {
				fprintf(terr, "VT n: %u, len: %u\n", nc, (unsigned)len);
				for (unsigned i = 0; i < nc; ++i) {
					const GdVt &p = vts[i];
					if (sr_variant) { // SR/map.c:701-716
						int32_t pos = p.first_target_loc + 1;
						if (p.str) pos -= (len - 1);
						fprintf(terr, "VT\t%s (len: %u)\t%d\t%c\t[%u, %u]\t%u\n", R.seq[p.chrom_id].name.c_str(), R.seq[p.chrom_id].len, pos, "+-"[p.str], p.first_query_loc, p.last_query_loc, p.score);
					} else
						fprintf(terr, "VT\t%s (len: %u)\t[%u, %u]\t%c\t[%u, %u]\t%u\n", R.seq[p.chrom_id].name.c_str(), R.seq[p.chrom_id].len, (uint32_t)p.first_target_loc,
						        (uint32_t)p.last_target_loc, "+-"[p.str], p.first_query_loc, p.last_query_loc, p.score);
				}
			}
			if (nc > 0) {
				std::vector<GdCand> C(nc);
				for (unsigned i = 0; i < nc; ++i) C[i].v = vts[i];
				if (sr_variant) gd_sr_boxes(C, O, R, (uint32_t)len), nc = (unsigned)C.size();
				else gd_lr_link_and_boxes(C, O, R, (uint32_t)len);
				if (stats) { // boxes the batch planner of the GPU path refuses (map_pipeline.hip.h: the reference's behaviour is undefined there)
					bool bad = false;
					for (unsigned i = 0; i < nc; ++i) bad |= C[i].qlen == 0 || C[i].tlen == 0 || C[i].qlen > (uint32_t)len || C[i].qseq_off + C[i].qlen > (uint32_t)len || C[i].tlen > 8u * (uint32_t)len + 100000u;
					if (bad) { ++st_degenerate; fprintf(stderr, "[stats] degenerate box in read %s\n", qn[ri].c_str()); }
				}
				if (trace && !sr_variant) {
					fprintf(terr, "AVT n: %u, len: %u\n", nc, (unsigned)len);
					for (unsigned i = 0; i < nc; ++i) {
						const GdVt &p = C[i].v;
						fprintf(terr, "AVT\t%s (len: %u)\t[%u, %u]\t%c\t[%u, %u]\t%u\tc:%u\n", R.seq[p.chrom_id].name.c_str(), R.seq[p.chrom_id].len, (uint32_t)p.first_target_loc,
						        (uint32_t)p.last_target_loc, "+-"[p.str], p.first_query_loc, p.last_query_loc, p.score, (unsigned)C[i].concat);
					}
					for (unsigned i = 0; i < nc; ++i) { // the reference prints the box before it widens it for reads <= 300 bp, in strand coordinates (LR/map.c:1659-1675)
						const GdVt &p = C[i].v;
						fprintf(terr, "BE\t%s, [%u, %u[ (chrom_len: %u) -> '%c' [%u, %u[ (read_len: %u)\n", R.seq[C[i].target_id].name.c_str(), (uint32_t)p.first_target_loc, (uint32_t)p.last_target_loc,
						        R.seq[C[i].target_id].len, "+-"[p.str], p.str ? (unsigned)len - 1 - p.last_query_loc : p.first_query_loc, p.str ? (unsigned)len - 1 - p.first_query_loc : p.last_query_loc, (unsigned)len);
					}
				}
				std::vector<GdDpResult> dp(nc);
				std::vector<gdo_extz_t> ez(nc);
				std::vector<uint32_t> one(nc);
				for (unsigned i = 0; i < nc; ++i) {
					GdCand &c = C[i];
					memset(&ez[i], 0, sizeof(gdo_extz_t));
					std::vector<uint8_t> t((size_t)c.tlen + 16, 0);
					gd_getseq(R, c.target_id, c.target_start, c.target_end + 1, t.data());
					const uint8_t *q = (c.v.str ? rev.data() : enc.data()) + c.qseq_off;
					bool exact = false;
					if (c.exact_score != GD_NEG_INF_SCORE) exact = gdo_exact_match((int)c.qlen, q, (int)c.tlen, t.data()) != 0;
					if (exact) one[i] = c.qlen << 4, dp[i] = {c.exact_score, &one[i], 1};
					else {
						gdo_ksw_extd2((int)c.qlen, q, (int)c.tlen, t.data(), 5, mat, (int8_t)O.q, (int8_t)O.e, (int8_t)O.q2, (int8_t)O.e2, dp_bw, -1, 0, GDO_EZ_APPROX_MAX | GDO_EZ_AVX512_SC, &ez[i]);
						dp[i] = {ez[i].score, ez[i].cigar, ez[i].n_cigar};
					}
					if (trace && !sr_variant) fprintf(terr, "AL_SCORE: %d\n", dp[i].score);
				}
				if (sr_variant) gd_sr_finish(C, dp, O, R, (uint32_t)len, enc.data(), rev.data(), regs);
				else gd_lr_finish(C, dp, O, R, (uint32_t)len, enc.data(), rev.data(), regs, terr);
				for (unsigned i = 0; i < nc; ++i) free(ez[i].cigar);
			}
		}
		const char *qual = qq[ri].size() == seq.size() && !qq[ri].empty() ? qq[ri].c_str() : nullptr;
		if (!regs.empty()) {
			for (size_t j = 0; j < regs.size(); ++j) {
				if ((O.flag & GD_F_NO_PRINT_2ND) && regs[j].id != regs[j].parent) continue;
				out.clear();
				if (out_sam) gd_write_sam(out, R, qn[ri].c_str(), seq.c_str(), qual, len, regs, (int)j, O.flag);
				else gd_write_paf(out, R, qn[ri].c_str(), len, regs, (int)j, O.flag);
				sam_of[ri] += out, sam_of[ri] += '\n';
			}
		} else if (out_sam || (O.flag & GD_F_PAF_NO_HIT)) { // LR/map.c:2176-2179
			out.clear();
			if (out_sam) gd_write_sam(out, R, qn[ri].c_str(), seq.c_str(), qual, len, regs, -1, O.flag);
			else gd_write_paf(out, R, qn[ri].c_str(), len, regs, -1, O.flag);
			sam_of[ri] += out, sam_of[ri] += '\n';
		}
		if (terr) {
			fclose(terr);
			trace_of[ri].assign(tbuf, tlen_);
			free(tbuf);
		}
	};
	std::atomic<size_t> next_read{0};
	std::vector<std::thread> workers;
	for (int t = 0; t < n_threads; ++t)
		workers.emplace_back([&] { for (size_t ri; (ri = next_read.fetch_add(1)) < qs.size();) map_one(ri); });
	for (auto &w : workers) w.join();
	if (stats) {
		uint64_t over_mid = 0, over_max = 0, max_cnt = 0, multi = 0;
		for (uint64_t s_ = 0; s_ < (1ull << V.tbits); ++s_)
			if (V.tkey[s_] != UINT64_MAX) {
				const uint32_t n = (uint32_t)V.tval[s_];
				over_mid += (int64_t)n > O.mid_occ, over_max += (int64_t)n > O.max_max_occ, multi += n > 1, max_cnt = n > max_cnt ? n : max_cnt;
			}
		fprintf(stderr, "[stats] index: keys=%llu multi=%llu keys>mid_occ(%d)=%llu keys>max_max_occ(%d)=%llu max_count=%llu\n", (unsigned long long)I.n_keys, (unsigned long long)multi,
		        O.mid_occ, (unsigned long long)over_mid, O.max_max_occ, (unsigned long long)over_max, (unsigned long long)max_cnt);
		fprintf(stderr, "[stats] reads=%zu mz_flt_drops=%ld (minimizers dropped %ld) high_occ_streak=%ld rescue=%ld heap_replace=%ld over_max_max_occ=%ld strand>4096hits=%ld max_strand_hits=%ld degenerate_box_reads=%ld\n",
		        qs.size(), st_mzflt.load(), st_mzflt_dropped.load(), st_high.load(), st_flt_rescued.load(), st_heap_replace.load(), st_over_max.load(), st_big_strand.load(), st_max_hits.load(), st_degenerate.load());
	}
	for (size_t ri = 0; ri < qs.size(); ++ri) {
		fputs(trace_of[ri].c_str(), stderr);
		fputs(sam_of[ri].c_str(), stdout);
	}
	return 0;
}
