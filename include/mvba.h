/* mvba.h -- C ABI of libmvba.so: MI355X-native bundle adjustment + factorization SVD.
 *
 * The reference (takah29/3d-reconstruction-from-multi-view-exp) has no FFI layer;
 * the surface it defines is Python:
 *     lib/bundle_adjustment.py:10-206   class BundleAdjuster  (ctor / optimize / get_log)
 *     lib/factorization.py:5-15         factorization_method(W, n_rank)
 * This header is what a ctypes binding underneath those two call sites binds
 * (INTEGRATION.md shows the stub).  The Levenberg-Marquardt control flow
 * (damping schedule, strict accept test, stop rule, log, print; ref :100-195)
 * stays in Python so it is bit-for-bit the reference's; everything that touches
 * observations, points or the reduced camera system is behind these calls.
 *
 * Conventions: every pointer is HOST memory, C-contiguous, little-endian;
 * doubles are IEEE binary64; no struct is passed by value; every function
 * returns an int status (0 = MVBA_OK) and mvba_last_error() gives the
 * thread-local message.  A handle is not thread-safe: one host thread per handle.
 * State (points, cameras) lives in the NORMALISED frame of ref :208-240; the
 * normalise / denormalise transforms are host-side NumPy in lib/bundle_adjustment.py.
 */
#ifndef MVBA_H
#define MVBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVBA_OK 0
#define MVBA_ERR_BADARG 1   /* -> ValueError              (ref :27-28, :231-232)              */
#define MVBA_ERR_SINGULAR 2 /* -> numpy.linalg.LinAlgError (ref :128 batched inv, :146 solve) */
#define MVBA_ERR_HIP 3      /* -> RuntimeError                                                */
#define MVBA_ERR_RCCL 4     /* -> RuntimeError                                                */
#define MVBA_ERR_STATE 5    /* call order violated (e.g. try_step before linearize)           */

typedef struct mvba_handle mvba_handle;

/* Observation list, CSR by point (replaces the dense x (N,m,2) + bool mask of
 * ref :37, :56-60).  In a point-sharded job n_points / n_obs / pt_ptr describe
 * THIS rank's points; n_images is global (cameras are replicated). */
typedef struct mvba_problem {
  int64_t n_points;       /* points held by this handle                          */
  int64_t n_obs;          /* = pt_ptr[n_points]                                  */
  int32_t n_images;       /* cameras (global)                                    */
  int32_t gauge_axis;     /* 0: "x-right_z-forward" (drops param 12), 1: "x-up_z-forward" (drops 13); ref :62-72 */
  const int64_t *pt_ptr;  /* [n_points+1] offsets into cam_idx / xy              */
  const int32_t *cam_idx; /* [n_obs] camera of each observation, ascending within a point */
  const double *xy;       /* [n_obs][2] observed image coordinates (xy_layout 0) */
  double f0;              /* ref :50                                             */
  int32_t device;         /* HIP device ordinal, -1 = current device             */
  int32_t xy_layout;      /* 0: xy in observation order.  1: xy as image planes [n_images][n_points][2] -- the memory of the
                           * reference caller's np.stack(x_list) (euclidiean_reconstruction.py:50) -- legal only when every point
                           * is observed in every image (n_obs = n_points * n_images; pt_ptr / cam_idx as always); the
                           * observation order is formed on the device instead of by a strided host copy               */
} mvba_problem;

/* Kernel ids for mvba_stats (names via mvba_kernel_name). */
enum {
  MVBA_K_RESID_JAC = 0, /* K1 residual + 2x3 / 2x9 Jacobians   (ref :291-427)            */
  MVBA_K_POINT_BLOCKS,  /* K2 E_a, dP_a (ref :429-469, :519-556): fused into K1, always 0 */
  MVBA_K_POINT_INV,     /* K3a damped 3x3 inverse, E^-1 dP      (ref :120-128)            */
  MVBA_K_SCHUR,         /* K3 A = G^ - sum F^T E^-1 F, b        (ref :132-143, :471-517, :618-664) */
  MVBA_K_ALLREDUCE,     /* C1 RCCL all-reduce of [A|b]                                    */
  MVBA_K_SOLVE,         /* K4 gauge strip + dense solve         (ref :146)                */
  MVBA_K_BACKSUB_COST,  /* K5+K6 dX, trial state, trial cost    (ref :152-162, :260-281, :666-677) */
  MVBA_K_COST,          /* residual-only cost pass              (ref :85-87)              */
  MVBA_K_COUNT
};

typedef struct mvba_stats {
  double ms[16];        /* accumulated device time per kernel id (hipEvents on the library's stream) */
  int64_t launches[16]; /* number of timed launches per kernel id                */
  int64_t n_linearize, n_try_step, n_commit;
  int64_t n_lu_fallback; /* solves that left the Cholesky path for LU with partial pivoting */
  int64_t n_barrier_fallback; /* solves redone with one launch per super-block because a wait of the persistent
                                 back-substitution timed out (its grid was not co-resident)                        */
} mvba_stats;

const char *mvba_version(void);
const char *mvba_last_error(void);
const char *mvba_kernel_name(int32_t kernel_id);
int mvba_device_count(int32_t *count);

/* Copies the observation list to the device and builds the camera-major index. */
int mvba_create(const mvba_problem *problem, mvba_handle **out);
void mvba_destroy(mvba_handle *h);

/* Committed state, normalised frame: X [n_points][3], f [m], u [m][2], t [m][3],
 * R [m][3][3] row-major with COLUMNS = camera axes (ref :40-48).               */
int mvba_set_params(mvba_handle *h, const double *X, const double *f, const double *u,
                    const double *t, const double *R);
int mvba_get_params(mvba_handle *h, double *X, double *f, double *u, double *t, double *R);

/* The way back to the caller's frame, on the device and in place (ref :242-258, applied by the
 * reference's optimize() before it returns, :198-200): X <- scale X R0^T + t0, t likewise,
 * R <- R0 R on the committed state.  R0 [3][3] row-major, t0 [3]. */
int mvba_apply_similarity(mvba_handle *h, const double *R0, const double *t0, double scale);

/* E = sum over observations of |e|^2 at the committed state (ref :666-677). */
int mvba_cost(mvba_handle *h, double *E);
/* K1+K2 at the committed state (ref :103-116). */
int mvba_linearize(mvba_handle *h);
/* One LM trial with damping c (ref :118-162): K3a,K3,(C1),K4,K5,K6; the trial
 * state stays on the device.  *E_trial is the job-wide cost at the trial state. */
int mvba_try_step(mvba_handle *h, double c, double *E_trial);
/* trial -> committed (ref :169-173). */
int mvba_commit(mvba_handle *h);

/* The debug log of the reference's optimize(is_debug=True) (ref :89-98, :175-183: a copy of X, R, t per outer
 * iteration, normalised frame; read back by get_log(), :204-206).  mvba_snapshot appends the COMMITTED state to
 * a log kept in device memory -- one device-to-device copy on the engine's stream, nothing crosses PCIe and the
 * host does not wait; mvba_snapshot_read fetches entry i (what get_log() does, once, afterwards);
 * mvba_snapshot_clear empties the log (the reference clears it at the start of every optimize, :90) and keeps
 * its memory for the next run; mvba_snapshot_restore makes entry i the committed state again (mvba_set_params
 * from device memory: linearisation and trial become void; nothing crosses PCIe). */
int mvba_snapshot(mvba_handle *h);
int mvba_snapshot_count(mvba_handle *h, int64_t *n);
int mvba_snapshot_read(mvba_handle *h, int64_t i, double *X, double *f, double *u, double *t, double *R);
int mvba_snapshot_clear(mvba_handle *h);
int mvba_snapshot_restore(mvba_handle *h, int64_t i);

/* Per-kernel device timing (hipEvents on the engine's stream); off by default.  enabled = 1: every phase;
 * 2: the Schur (K3) and residual-Jacobian (K1) kernels only -- each timed phase is two marker packets on the stream. */
int mvba_set_profiling(mvba_handle *h, int32_t enabled);
int mvba_get_stats(mvba_handle *h, mvba_stats *out);
int mvba_reset_stats(mvba_handle *h);

/* Sizes of the Schur index built at create and what the communicator runs on (bench.py prices the
 * kernels with them): out[0] (point, camera pair) items incl. diagonal pairs, out[1] off-diagonal
 * items, out[2] units (wave runs / slot lists), out[3] Schur kernel form: 0 = camera strips (round 1),
 * 1 = pair-major units (round 2), 2 = slot-resident (round 3), 3 = dense visibility (round 5: at most 21 cameras and at least
 * 60 % of the (point, camera) pairs observed: no pair index, out[0..2] = 0) in bits 0..7, the slot form's rounds (camera-group pairs swept one
 * after the other inside the launch) in bits 8..31 and camera groups in bits 32.., out[4] ncclGetVersion() of the librccl
 * actually loaded (0 without a communicator), out[5] the NCCL_VERSION_CODE the library was compiled
 * against, out[6] ranks, out[7] slot form: step-major item rows including the padding rows. */
int mvba_get_info(mvba_handle *h, int64_t *out8);

/* Point-sharded multi-GPU (one process per GPU): rank 0 makes an id, the host
 * side ships its 128 bytes to the other ranks, every rank calls comm_init.
 * Afterwards try_step all-reduces the partial reduced system [A|b] over RCCL and
 * cost / try_step return the sum of the ranks' costs taken in rank order.      */
int mvba_comm_unique_id(void *id128);
int mvba_comm_init(mvba_handle *h, const void *id128, int32_t rank, int32_t n_ranks);

/* The same sharded job over a HOST-STAGED transport instead of RCCL: the library copies the packed
 * [A|b] (and the 16-byte cost/status record) to the host and calls `fn(user, buf, n)`, which must
 * sum `buf` in place over all ranks (e.g. a gloo / MPI all-reduce) and return 0.  For machines
 * without RCCL and for multi-process tests that share one GPU (RCCL refuses two ranks on one
 * device); control flow, rank-ordered cost sum and collective error behaviour are identical. */
typedef int (*mvba_host_allreduce_fn)(void *user, double *buf, int64_t n);
int mvba_comm_init_host(mvba_handle *h, int32_t rank, int32_t n_ranks, mvba_host_allreduce_fn fn, void *user);

/* Test hook: download an intermediate in canonical per-observation / per-point
 * row-major layout.  Returns the element count in *n (out may be NULL to query). */
enum {
  MVBA_BUF_RESIDUAL = 0, /* [n_obs][2]                                       */
  MVBA_BUF_JX,           /* [n_obs][2][3]                                    */
  MVBA_BUF_JC,           /* [n_obs][2][9]                                    */
  MVBA_BUF_E,            /* [n_points][6]  xx,xy,xz,yy,yz,zz  (undamped)     */
  MVBA_BUF_DP,           /* [n_points][3]                                    */
  MVBA_BUF_A_FULL,       /* [9m][9m] symmetric, before gauge removal         */
  MVBA_BUF_B_FULL,       /* [9m]                                             */
  MVBA_BUF_DXI,          /* [9m] with zeros at the gauge slots               */
  MVBA_BUF_DX,           /* [n_points][3]                                    */
  MVBA_BUF_TRIAL_X,      /* [n_points][3]                                    */
  MVBA_BUF_TRIAL_CAM,    /* [m][15]  f,u,v,t[3],R[9]                         */
  MVBA_BUF_INDEX_K,      /* the Schur index as the kernel reads it: k-side observation of every item row */
  MVBA_BUF_INDEX_L,      /*   l-side observation                                                          */
  MVBA_BUF_INDEX_A,      /*   point (slot form: step-major rows incl. padding; unit form: pair-major)      */
  MVBA_BUF_INDEX_SEG     /*   slot form: pacing table [waves][segments]                                   */
};
int mvba_debug_read(mvba_handle *h, int32_t which, double *out, int64_t capacity, int64_t *n);

/* Pinhole projection of an observation list on the device: xy[o] = inhomogeneous
 * K_k [R_k^T | -R_k^T t_k] [X_a; 1] (ref lib/camera.py:13-14, :30-34, :74-81 -- the scene side of
 * BA: synthetic observations before it, re-projection after it).  K, R [n_images][3][3] row-major
 * (R: columns = camera axes), t [n_images][3].  pt_ptr / cam_idx as in mvba_problem; pt_ptr == NULL
 * means the dense grid (n_obs = n_points * n_images, observation = point * n_images + camera: the
 * reference's calc_projected_points, transposed).  xy [n_obs][2] is written. */
int mvba_project(const double *X, int64_t n_points, const double *K, const double *R, const double *t, int32_t n_images,
                 const int64_t *pt_ptr, const int32_t *cam_idx, int64_t n_obs, double *xy, int32_t device);

/* Host-only check of the per-observation math the kernels use (no GPU needed):
 * cam15 = f,u,v,t[3],R[9]; out = e[2], JX[6], JC[18].                          */
int mvba_host_obs_math(const double *X3, const double *cam15, const double *xy2, double f0,
                       double *out26);

/* ---- factorization (ref lib/factorization.py:5-15) ---------------------------
 * Wt: the measurement matrix as its callers hold it, row-major [n_rows][n_cols]
 * with n_rows = points (tall) and n_cols = 2m or 3m; the reference's W is Wt^T
 * (perspective_camera_calibration.py:533, affine_camera_calibration.py:236).
 * dtype: 0 = float32, 1 = float64 (outputs have the input dtype, ref quirk B.10).
 * center != 0: subtract the column means of Wt first (= the row means of W the
 * affine callers remove, affine_camera_calibration.py:224-240); means [n_cols]
 * receives them (may be NULL).
 * Outputs: M [n_cols][n_rank] (= U[:, :r]), sigma [n_cols] (singular values,
 * descending), S [n_rank][n_rows] (= diag(sigma[:r]) Vt[:r] = M^T W).  Thin: Vt is
 * never formed.  Sign convention: the largest-magnitude component of every column
 * of M is positive (LAPACK's signs are not a rule one can restate).
 * Two routes.  n_cols <= 64, or n_rank > 16 with n_cols <= 256: Gram matrix + Jacobi, n_rank any 1 .. n_cols, sigma = every
 * singular value.  Otherwise (65 .. 12288 columns = three rows per image at the engine's 4096 cameras, n_rank <= 16; MVBA_ERR_BADARG
 * beyond either): block power iteration with Rayleigh-Ritz on W^T W applied implicitly, 32 vectors wide, two passes over W per
 * iteration, no n_cols x n_cols matrix; sigma[0..31] = the block's Ritz values (the leading n_rank converged), sigma[32..] = NaN;
 * MVBA_ERR_SINGULAR ("SVD did not converge", LAPACK's own failure) after 2000 iterations.
 * Accuracy: float32 data -> one Gram pass accumulated in fp64 (nothing is lost: eps32 >> eps64 *
 * cond^2); float64 data -> a second, preconditioned pass so that small singular values are good to
 * ~eps64 * sigma_1 like LAPACK's gesdd, not to sqrt(eps64) * sigma_1 (see csrc/mvsvd.hip).
 * timings_ms (may be NULL) [6]: H2D, means + Gram, Jacobi, projection (device ms), sweeps,
 * refinement pass (0 for float32).  Block route: H2D, means, the iteration, S out of the last product, iterations, final pass. */
int mvsvd_factorize(const void *Wt, int64_t n_rows, int32_t n_cols, int32_t dtype, int32_t n_rank,
                    int32_t center, void *M, void *sigma, void *S, void *means, double *timings_ms,
                    int32_t device);

/* Workspace form for repeated factorizations (the projective-depth loops call the SVD 50-200
 * times, ref perspective_camera_calibration.py:61-144, :147-235): create once (device buffers
 * for up to max_rows x n_cols, stream, events), load a matrix (the only host-to-device copy),
 * run any number of factorizations on the resident matrix (different n_rank / center), destroy. */
typedef struct mvsvd_handle mvsvd_handle;
int mvsvd_create(int64_t max_rows, int32_t n_cols, int32_t dtype, int32_t device, mvsvd_handle **out);
int mvsvd_load(mvsvd_handle *h, const void *Wt, int64_t n_rows);
/* The same matrix put together on the device from the images' own arrays -- what the reference's callers hold
 * (ref lib/affine_camera_calibration.py:224-240: W = np.hstack(data_list).T; the hstack alone is 0.3 s of strided host writes at
 * 5 M points x 12 images, twenty times the upload + factorisation): xy[k] = image k's coordinates [n_rows][2], float32
 * (src_dtype 0) or float64 (1), converted to the workspace's dtype; W^T[i][2k .. 2k+1] = xy[k][i]; n_cols must be 2 n_images. */
int mvsvd_load_images(mvsvd_handle *h, const void *const *xy, int32_t n_images, int64_t n_rows, int32_t src_dtype);
int mvsvd_run(mvsvd_handle *h, int32_t n_rank, int32_t center, void *M, void *sigma, void *S, void *means,
              double *timings_ms);

/* The projective-depth loops (ref perspective_camera_calibration.py:61-144, :147-235) factorise, 50-200 times,
 * the SAME homogeneous observations X re-weighted by the current depths: W[a][g*group + c] = X[a][g*group + c] *
 * z[a][g] * s with s = 1 / |row a of X o z| (norm 1: every row -- point -- to unit length, ref :86-87), s = 1 / sum of
 * the group's (X o z)^2 over all rows (norm 2: every column group -- image -- divided by its squared Frobenius
 * norm, ref :170-172) or s = 1 (norm 0).  mvsvd_load_base uploads X (n_rows x n_cols, the workspace's dtype) once;
 * every mvsvd_run_scaled uploads only z (n_rows x n_cols / group), forms W on the device and runs the factorisation
 * (no centring).  timings_ms[0] is then the upload of z.  z == NULL: the depths a device depth loop (below) left in the workspace,
 * same grouping -- nothing is uploaded: the final factorisation of perspective_self_calibration (ref :531-533) without W on the host. */
int mvsvd_load_base(mvsvd_handle *h, const void *X, int64_t n_rows);
/* The same base assembled on the device from the images' own arrays (ref lib/perspective_camera_calibration.py:34-40,
 * _create_data_matrix: 0.11 s of strided host writes at 1 M points x 12 images, and a third more bytes over PCIe):
 * xy[k] = image k's pixel coordinates [n_rows][2] (doubles), base[i][3k .. 3k+2] = (x / f0, y / f0, 1) in the workspace's
 * dtype; n_cols of the workspace must be 3 n_images. */
int mvsvd_load_base_images(mvsvd_handle *h, const double *const *xy, int32_t n_images, int64_t n_rows, double f0);
int mvsvd_run_scaled(mvsvd_handle *h, const void *z, int32_t group, int32_t norm, int32_t n_rank, void *M, void *sigma, void *S,
                     double *timings_ms);

/* The depth UPDATE of those loops on the device too (ref perspective_camera_calibration.py:93-129 primary, :182-224
 * dual): the depths z live in the workspace, and one mvsvd_depth_step is one whole iteration of the reference's loop --
 * re-weight the resident X by z and normalise (norm = method), rank-4 factorisation, then per point (method 1, primary)
 * the dominant eigenvector of the m x m matrix of :99-107 from its 4 x 4 companion, or per image (method 2, dual) that
 * of the N x N matrix of :188-205 from its 12 x 12 companion (O(N) memory), the sign rules of :121 / :217,
 * z <- xi / |x| (:124 / :220) and the reprojection error of :43-58 into *E.  Per iteration 8 bytes cross PCIe.
 * mvsvd_depth_begin (after mvsvd_load_base; group must be 3: homogeneous image coordinates, n_cols = 3 m) sets
 * z = 1 (:75 / :160); mvsvd_depth_read downloads the depths [n_rows][m] (the workspace's dtype), once, at the end.
 * timings_ms (may be NULL) [6] as in mvsvd_run, except slot 0: device ms of the depth-update kernels.
 * MVBA_ERR_SINGULAR: the re-weighted matrix has rank < 4. */
int mvsvd_depth_begin(mvsvd_handle *h, int32_t group);
int mvsvd_depth_step(mvsvd_handle *h, int32_t method, double f0, double *E, double *timings_ms);
int mvsvd_depth_read(mvsvd_handle *h, void *z);
void mvsvd_destroy(mvsvd_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* MVBA_H */
