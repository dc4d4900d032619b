/* oracle/gdo_lchain.h -- CPU ORACLE (test infrastructure, NOT product code): see gdo_lchain.c */
#ifndef GDO_LCHAIN_H
#define GDO_LCHAIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void gdo_radix_sort_128x(uint64_t *a, int64_t n);
uint64_t *gdo_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc, float chn_pen_gap,
                        float chn_pen_skip, int is_cdna, int n_seg, int64_t n, const uint64_t *a, int *n_u, uint64_t **u, int32_t *f_out,
                        int64_t *p_out);
#ifdef __cplusplus
}
#endif
#endif
