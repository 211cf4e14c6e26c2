// libmvba.so -- tall-skinny SVD for the factorization init (ref lib/factorization.py:5-15).
#include "mvba_common.h"

extern "C" int mvsvd_factorize(const void *Wt, int64_t n_rows, int32_t n_cols, int32_t dtype, int32_t n_rank,
                               void *M, void *sigma, void *S, int32_t device) {
  (void)Wt; (void)n_rows; (void)n_cols; (void)dtype; (void)n_rank; (void)M; (void)sigma; (void)S; (void)device;
  return mvba::fail(MVBA_ERR_STATE, "mvsvd_factorize: not implemented yet");
}
