/* fiat_amd -- MI355X-native batched finite-element tabulator: C ABI.
 *
 * The reference (firedrakeproject/fiat) has no FFI layer: the boundary of its
 * tabulate() hot path is the Python object API.  Each entry point below names
 * the reference interface it stands behind (paths relative to the reference
 * root).  The Python facade in fiat_amd/ binds these with ctypes and keeps the
 * reference's class/method signatures; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - all arithmetic and all bulk arrays are IEEE double, row-major, dense;
 *   - pointers documented "device" are HBM pointers valid on the context's
 *     GPU (16-byte aligned); "host" pointers are ordinary memory;
 *   - the library never frees or retains caller memory;
 *   - every function returns FX_OK (0) or a negative FX_E* code and records a
 *     message retrievable with fx_last_error() (thread local);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *     work is enqueued asynchronously on it, nothing synchronises unless said;
 *   - there is NO CPU fallback: with no usable gfx950 device fx_ctx_create fails.
 */
#ifndef FIAT_AMD_H
#define FIAT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_OK 0
#define FX_EINVAL (-1)    /* ValueError in the reference            */
#define FX_ENOTIMPL (-2)  /* NotImplementedError in the reference   */
#define FX_ESINGULAR (-3) /* numpy.linalg.LinAlgError (finite_element.py:151-156) */
#define FX_EHIP (-4)      /* HIP runtime failure                    */
#define FX_ENOMEM (-5)

/* expansion-set variants (FIAT/expansions.py:174-177) */
#define FX_VARIANT_DEFAULT 0 /* variant=None : orthonormal Dubiner  */
#define FX_VARIANT_BUBBLE 1  /* "bubble": integrated Jacobi + C0_basis */
#define FX_VARIANT_DUAL 2    /* "dual" */

typedef struct fx_ctx fx_ctx;
typedef struct fx_element fx_element;
typedef struct fx_line_element fx_line_element;

const char* fx_last_error(void);
int fx_abi_version(void);

/* One context per GPU (one process per GPU in multi-GPU runs). */
int fx_ctx_create(int device_id, fx_ctx** ctx);
int fx_ctx_destroy(fx_ctx* ctx);
/* Device facts used by the benchmark: CU count and bytes of LDS per CU. */
int fx_ctx_info(fx_ctx* ctx, int* num_cu, int* lds_bytes_per_cu, char* name, int name_len);

/* Kernel-selection policy of a context (no reference counterpart: FIAT has one NumPy path,
 * FIAT/expansions.py:449-490).  Several kernels can serve a request shape; fx_tabulate_batch picks the
 * fastest registered one.  The bits below take a kernel family out of the selection (or opt into an A/B
 * partner) so that parity tests and tools can reach every kernel through the same entry points.  Results
 * are identical to the stated tolerance under every policy; 0 is the default.  Not read from the
 * environment: launch paths make no getenv calls. */
#define FX_POLICY_NO_FIXED (1u << 0)       /* shape-specialised paired / K-streamed / LDS-image kernels */
#define FX_POLICY_NO_SMALL (1u << 1)       /* lane-local low-order kernel */
#define FX_POLICY_NO_STACKED (1u << 2)     /* stacked-matrix kernel */
#define FX_POLICY_NO_COOP (1u << 3)        /* cooperative producer/consumer kernel */
#define FX_POLICY_STACKED_SMALL (1u << 4)  /* opt in: register-resident stacked instances on the small shapes */
#define FX_POLICY_NO_STACKED_MIX (1u << 5) /* chain rule of per-request cells as a second pass, not in the kernel */
#define FX_POLICY_NO_SHARED_WAVE (1u << 6) /* fx_tabulate_batch_shared: wave-per-request kernel */
#define FX_POLICY_NO_SHARED_REG (1u << 7)  /* fx_tabulate_batch_shared: register-resident kernel */
#define FX_POLICY_NO_MACRO_SMALL (1u << 8) /* fx_macro_tabulate_batch: lane-local kernel */
#define FX_POLICY_KERNEL_IMAGE (1u << 9)   /* shape-specialised family: LDS-image variant */
#define FX_POLICY_KERNEL_STREAM (1u << 10) /* shape-specialised family: one request per wave, K-streamed */
#define FX_POLICY_NO_WG (1u << 11)         /* request-per-workgroup kernel (rules of 49..128 points): point chunks instead */
#define FX_POLICY_WG_SMALL (1u << 12)      /* opt in: the request-per-workgroup kernel with several small requests per workgroup (<= 64 points) */
#define FX_POLICY_NO_SMALL_VALUES (1u << 13) /* lane-local kernel: not for the values-only P5 triangles / P3 tetrahedra it took over in round 4 */
#define FX_POLICY_ALL ((1u << 14) - 1)
int fx_ctx_set_policy(fx_ctx* ctx, unsigned flags);
int fx_ctx_get_policy(const fx_ctx* ctx, unsigned* flags);

/* ---- simplex elements ------------------------------------------------------
 * A polynomial set over the Dubiner expansion set of one simplex cell:
 *   coeffs[ndof][vdim][nexp]  (FIAT/polynomial_set.py:42-66, PolynomialSet),
 *   nexp = C(n+sd, sd).  coeffs == NULL means the identity (ONPolynomialSet,
 *   polynomial_set.py:110-134): tabulation then returns the raw expansion set,
 *   i.e. ExpansionSet._tabulate (expansions.py:449-490).
 * verts: host, (sd+1)*sd cell vertices, or NULL for the UFC simplex
 *   (reference_element.py:1006-1152).  scale: first-member scale
 *   (expansions.py:371-373, 386-399); pass <= 0 for the default
 *   sqrt(1/vol(default simplex)).
 */
int fx_element_create(fx_ctx* ctx, int sd, int n, int variant, double scale,
                      const double* verts, int ndof, int vdim,
                      const double* coeffs, fx_element** elem);
int fx_element_destroy(fx_element* elem);
/* Replace the coefficient tensor (after the Vandermonde solve). Host pointer. */
int fx_element_set_coeffs(fx_element* elem, int ndof, int vdim, const double* coeffs);
int fx_element_dims(const fx_element* elem, int* sd, int* n, int* nexp, int* ndof, int* vdim);

/* Number of derivative multi-indices of total order <= order: C(sd+order, sd)
 * (keys of the dict built at expansions.py:427-432, in mis() order). */
int fx_num_tables(int sd, int order);

/* THE HOT ENTRY.  Batched CiarletElement.tabulate (finite_element.py:181-197
 * -> polynomial_set.py:68-72 -> expansions.py:411-447,140-267,270-322).
 *   pts   device [nreq][npts][sd]   points in the coordinates of each request's cell
 *   verts device [nreq][sd+1][sd]   per-request cell vertices, or NULL: every
 *                                    request uses the element's own cell
 *   out   device [nreq][ntab][ndof][vdim][npts],  ntab = fx_num_tables(sd, order);
 *         table t of request r is exactly tabulate(order, pts[r])[alpha_t].
 * order <= 2 by the recurrence (FIAT/expansions.py:140-267); orders 3..8 through differentiation matrices
 * (expansions.py:438-446, 577-599), with per-request cells up to order 4.  Derivatives are with respect to the
 * caller's coordinates.  Stream-asynchronous, with one exception: the FIRST call of an element at an order > 2 builds
 * its differentiation matrices (null-stream launches, synchronous copies) and thereby synchronises the device once. */
int fx_tabulate_batch(fx_ctx* ctx, const fx_element* elem, int order,
                      int64_t nreq, int npts, const double* pts,
                      const double* verts, double* out, void* stream);

/* Same, host pointers, synchronous (stages through device memory). */
int fx_tabulate_batch_host(fx_ctx* ctx, const fx_element* elem, int order,
                           int64_t nreq, int npts, const double* pts,
                           const double* verts, double* out);

/* ---- Vandermonde / Riesz assembly and solve -----------------------------------
 * DualSet.to_riesz (dual_set.py:150-172): mat[i][v][k] = sum_q wts[i][v][q] *
 * expvals[k][q], where expvals is the expansion set tabulated at the union of
 * functional points (use fx_tabulate_batch with identity coeffs, order 0).
 * All device pointers. */
int fx_riesz_assemble(fx_ctx* ctx, int nrows, int nq, int nexp,
                      const double* wts, const double* expvals, double* mat, void* stream);

/* CiarletElement.__init__ (finite_element.py:141-159), batched:
 *   V = A B^T ; X = solve(V^T, B)   with A = dualmat[nsys][ndof][m], B =
 *   prime coeffs [nsys][ndof][m] (m = vdim*nexp).  X device [nsys][ndof][m],
 *   Vout (optional, may be NULL) device [nsys][ndof][ndof], info device
 *   int[nsys]: 0 ok, >0 zero pivot (singular) at that column.  Partial-pivot LU. */
int fx_vandermonde_solve_batch(fx_ctx* ctx, int64_t nsys, int ndof, int m,
                               const double* A, const double* B, double* X,
                               double* Vout, int* info, void* stream);

/* ---- 1-D Lagrange by barycentric interpolation and tensor products -----------
 * LagrangeLineExpansionSet / barycentric_interpolation / make_dmat
 * (barycentric_interpolation.py:22-93).  nodes: host [nn], 1 <= nn <= 256 (FX_ENOTIMPL beyond); the tensor-product and
 * prism entries below take factors of up to 16 nodes (their kernels keep a factor's basis in registers) and return
 * FX_ENOTIMPL for more -- fx_line_tabulate_batch serves any element this call creates. */
int fx_line_element_create(fx_ctx* ctx, int nn, const double* nodes, fx_line_element** elem);
int fx_line_element_destroy(fx_line_element* elem);

/* Batched 1-D tabulate: pts device [nreq][npts], out device [nreq][order+1][nn][npts]. */
int fx_line_tabulate_batch(fx_ctx* ctx, const fx_line_element* elem, int order,
                           int64_t nreq, int npts, const double* pts, double* out, void* stream);

/* TensorProductElement.tabulate, scalar factors (tensor_product.py:231-292),
 * nested left to right over nf <= 3 interval factors (quad / hex).
 *   pts device [nreq][npts][nf];  out device [nreq][ntab][prod nn_f][npts],
 *   ntab = fx_num_tables(nf, order), basis index a*nB + b (row-major over factors). */
int fx_tensor_tabulate_batch(fx_ctx* ctx, int nf, const fx_line_element* const* factors,
                             int order, int64_t nreq, int npts, const double* pts,
                             double* out, void* stream);

/* Sum-factorised variant: per-request 1-D coordinates of a tensor grid,
 *   grid device [nreq][nf][q]; points are the q^nf grid (x fastest last, as
 *   make_tensor_product_quadrature, quadrature.py:258-268); same out layout
 *   with npts = q^nf. */
/* TensorProductElement.tabulate for a prism, fused: `tri` an element on a triangle (scalar- or vector-valued), `line` a 1-D
 * Lagrange element (FIAT/tensor_product.py:231-317: table alpha = (alpha_A, alpha_B) is the per-point product of the factors'
 * tables, basis function (a, b) -> a * dim(B) + b, the value component rides on the vector-valued factor).
 * pts[nreq][npts][3] = (x, y | z), out[nreq][ntab][ndofA * ndofB][vdimA][npts], tables in mis(3, k) order.
 * Returns FX_ENOTIMPL for shapes without an instance (degree of the triangle factor 1..3, 1..4 nodes, order <= 2 [<= 1 for
 * degree 3 or 4 nodes], request <= 20 KB): the general route (two factor tabulations + fx_table_outer_batch) serves those. */
int fx_prism_tabulate_batch(fx_ctx* ctx, const fx_element* tri, const fx_line_element* line, int order, int64_t nreq,
                            int npts, const double* pts, double* out, void* stream);

int fx_tensor_tabulate_grid_batch(fx_ctx* ctx, int nf, const fx_line_element* const* factors,
                                  int order, int64_t nreq, int q, const double* grid,
                                  double* out, void* stream);

/* ---- plan introspection (host only, no GPU needed) -------------------------------
 * The device kernels never evaluate recurrence coefficients: the host flattens
 * dubiner_recurrence (expansions.py:140-267) into a table of steps
 *   m_dst = (A*fa - B*fb) * m_cur - C*fc * m_prv        (fa, fb, fc of `codim`)
 * with the per-codimension normalisation (expansions.py:251-266) folded into
 * (A, B, C).  These two calls expose the table and the C0_basis matrix
 * (expansions.py:270-322) so the CPU test-suite can check them against the oracle.
 *   ints[i] = {dst, cur, prv (-1: none), codim}, coefs[i] = {A, B, C}. */
int fx_plan_steps(int sd, int n, int variant, double scale, int cap, int* nsteps,
                  double* phi0, int* ints, double* coefs);
int fx_plan_c0_transform(int sd, int n, double* T /* host [nexp][nexp] */);
/* The cooperative kernel's schedule of the same recurrence (plan.hpp build_coop_plan): per
 * producer w its entries ints[w][i] = {level (-1: constant / zero row), seed, publish (K slot + 1,
 * 0: none), member, 0}, the entry ranges of the K-steps kstart[w][0..KS] and the K order
 * kperm[4*j + slot] = member (-1: zero row).  Host only, for the CPU test-suite. */
int fx_plan_coop(int sd, int n, int variant, double scale, int cap, int* KS, int* nentries,
                 int* ints, int* kstart, int cap_k, int* kperm);

/* ---- push-forward to physical cells ---------------------------
 * Reference: FiniteElement.mapping() (FIAT/finite_element.py:84-88) names the map, the
 * consumer applies it (finat/hdivcurl.py:95-191; checked in test/finat/test_point_evaluation.py:53-70).
 * fx_tabulate_batch with verts != NULL already returns derivatives with respect to the
 * PHYSICAL coordinates (affine pull-back: nothing left to do for FX_MAP_AFFINE).  For
 * vector-valued elements (value shape (sd,)) this call applies, in place and to every table,
 *   FX_MAP_COVARIANT_PIOLA      phi = J^{-T} Phi        (Nedelec)
 *   FX_MAP_CONTRAVARIANT_PIOLA  phi = J Phi / det J     (Raviart-Thomas)
 * with J = dx/dX between the element's own cell and the request's cell `verts[r]`.
 * out: the [nreq][ntab][ndof][sd][npts] device tensor fx_tabulate_batch produced with the
 * same verts.  (Round 1: a separate pass over the tables, not yet fused into the kernels.) */
#define FX_MAP_AFFINE 0
#define FX_MAP_COVARIANT_PIOLA 1
#define FX_MAP_CONTRAVARIANT_PIOLA 2
/* matrix-valued elements, value shape (sd, sd) (fx_pushforward_batch / fx_tabulate_batch_mapped only):
 *   FX_MAP_DOUBLE_COVARIANT_PIOLA      phi = J^{-T} Phi J^{-1}        (Regge)
 *   FX_MAP_DOUBLE_CONTRAVARIANT_PIOLA  phi = J Phi J^T / (det J)^2    (Hellan-Herrmann-Johnson) */
#define FX_MAP_DOUBLE_COVARIANT_PIOLA 3
#define FX_MAP_DOUBLE_CONTRAVARIANT_PIOLA 4
/*   FX_MAP_COVARIANT_CONTRAVARIANT_PIOLA  phi = J^{-T} Phi J^T / det J   (Gopalakrishnan-Lederer-Schoeberl;
 *   "covariant contravariant piola" of FIAT/gopalakrishnan_lederer_schoberl.py:67) */
#define FX_MAP_COVARIANT_CONTRAVARIANT_PIOLA 5
int fx_pushforward_batch(fx_ctx* ctx, const fx_element* elem, int mapping, int order,
                         int64_t nreq, int npts, const double* verts, double* out, void* stream);
/* fx_tabulate_batch followed by the push-forward `mapping`, in ONE kernel where the shape's kernel
 * fuses it into its output stage (N2 / RT2 tetrahedra, order 1), otherwise as two launches. */
int fx_tabulate_batch_mapped(fx_ctx* ctx, const fx_element* elem, int mapping, int order,
                             int64_t nreq, int npts, const double* pts, const double* verts,
                             double* out, void* stream);

/* ---- one reference point set in many cells (the quadrature-rule case) ---------------
 * What FIAT's consumers do around the reference (finat/fiat_elements.py:69: tabulate once on the
 * reference cell; TSFC-generated code pushes forward per cell) as one call: ref_pts[npts][sd] are
 * coordinates on the ELEMENT's own cell (device pointer), request r is the affine image of that
 * cell with vertices verts[r]; out[r] = the element's tables pushed forward to cell r with
 * `mapping` (FX_MAP_*): exactly what fx_tabulate_batch(points = F_r(ref_pts), verts) followed
 * by fx_pushforward_batch returns, without the per-request recurrence (second
 * variant of config 2; 8(f) ranks 1-2).  out: [nreq][ntab][ndof][vdim][npts] device tensor. */
int fx_tabulate_batch_shared(fx_ctx* ctx, const fx_element* elem, int mapping, int order,
                             int64_t nreq, int npts, const double* ref_pts, const double* verts,
                             double* out, void* stream);

/* ---- point production on the device ---------------------------
 * Collapsed Gauss-Jacobi rule with m points per direction on the simplex `verts`
 * ((sd+1) x sd HOST doubles; NULL = the UFC simplex): FIAT/quadrature.py
 * CollapsedQuadratureSimplexRule (:171-181) / GaussJacobiQuadratureLineRule (:96-110) =
 * create_quadrature(ref_el, 2m-1, "collapsed").  pts[m^sd][sd] and wts[m^sd] are DEVICE
 * buffers, points ordered with the last direction fastest; the points can be handed to
 * fx_tabulate_batch_shared as they are.  The Gauss-Jacobi nodes are Newton roots of the Jacobi
 * recurrence of FIAT/jacobi.py:47-74. */
int fx_collapsed_quadrature(fx_ctx* ctx, int sd, int m, const double* verts, double* pts,
                            double* wts, void* stream);

/* ---- the FInAT side of the boundary ---------------------------
 * What finat/fiat_elements.py:92-111 (FiatElement.basis_evaluation) checks about each FIAT table
 * before it becomes a GEM literal, for `ntables` device-resident tables [rows][npts] at once:
 * stats[t][0] = max |x| (a table of derivative order > degree must be zero, :104-108),
 * stats[t][1] = max (|x - x[.., 0]| - rtol |x[.., 0]|) (derivative order == degree on a simplex:
 * constant over the points, numpy.allclose(table, table[..., 0, None]), :96-103; compare with
 * numpy's atol).  NaN entries give NaN stats.  tables, stats[ntables][2]: device pointers. */
int fx_classify_tables(fx_ctx* ctx, int64_t ntables, int rows, int npts, double rtol,
                       const double* tables, double* stats, void* stream);
/* FIAT/finite_element.py:222-264 (entity_support_dofs; FInAT's twin finat/finiteelementbase.py:85-119): the squared
 * L2 norm of every basis function over a sub-entity, out[t][r] = sum_p weights[p] sum_c tables[t][r][c][p]^2 for
 * `ntables` device-resident order-0 tables [rows][vdim][npts] evaluated at the points of ONE quadrature rule
 * (weights[npts]) mapped onto the entities (fx_map_points + fx_tabulate_batch); a dof belongs to the support of
 * entity t when out[t][dof] > 1e-8.  tables, weights, out[ntables][rows]: device pointers. */
int fx_tables_squared_norm(fx_ctx* ctx, int64_t ntables, int rows, int vdim, int npts, const double* tables,
                           const double* weights, double* out, void* stream);
/* [ntables][rows][npts] -> [ntables][npts][rows]: the layout FInAT hands tables to generated kernels
 * in (finat/runtime_tabulated.py:79: point extents, then index_shape + value_shape).  in != out. */
int fx_tables_point_major(fx_ctx* ctx, int64_t ntables, int rows, int npts, const double* in,
                          double* out, void* stream);

/* Name of the device kernel fx_tabulate_batch would launch for this element and request shape
 * ("fxk::tabulate_simplex_stream", "..._fixed", "..._coop" or the generic "..._kernel"); has_verts != 0
 * stands for per-request cell geometry (bit 0; bit 1: append the registry instance of the stacked-matrix kernel,
 * "...stacked<sd,n,column tiles,requests per group,kind>", "+piola" when the instance applies the map of bits 2-3 --
 * FX_MAP_COVARIANT_PIOLA / FX_MAP_CONTRAVARIANT_PIOLA << 2 -- itself).  Lets benchmarks and tests name the kernel they measured
 * (no reference counterpart: FIAT has a single NumPy path, FIAT/expansions.py:449-490). */
int fx_plan_kernel(fx_ctx* ctx, const fx_element* elem, int order, int64_t nreq, int npts, int has_verts, char* name,
                   int name_len);

/* ---- macro elements: tabulation on a split cell (simplicial complex) ------------------
 * ExpansionSet._tabulate on a macro cell (FIAT/expansions.py:449-490): every point is binned to the
 * sub-cell(s) it lies in (compute_cell_point_map :771-811, l1 distance in rescaled barycentric
 * coordinates, reference_element.py:616-644,778-780, tolerance 1e-12), the expansion set of that
 * sub-cell is evaluated there (:411-447), divided by the multiplicity of the point where the binning
 * is not unique (:469-477), and scattered into the members cell_node_map[cell] (:479-490,
 * polynomial_cell_node_map :744-768); PolynomialSet.tabulate contracts with coeffs
 * (polynomial_set.py:68-72).  Binning is unique (first cell) for C0 sets ("bubble") at order 0 (:452).
 *   parent_verts  host [sd+1][sd]          the split simplex (NULL: UFC simplex)
 *   cell_verts    host [ncell][sd+1][sd]   vertices of the sub-cells, in the parent's coordinates
 *   cell_node_map host [ncell][nexp]       member of the complex for every member of a sub-cell,
 *                                          nexp = C(n+sd, sd); values in [0, nmacro)
 *   cell_scale    host [ncell] or NULL     per-cell factor on the expansion values (string scales
 *                                          "orthonormal" / "L2 piola" of get_scale :386-399)
 *   coeffs        host [ndof][vdim][nmacro] or NULL = identity (ndof*vdim == nmacro)
 * ncell <= 32.  fx_macro_tabulate_batch: same argument meaning and output layout as
 * fx_tabulate_batch; with per-request cells (verts) the points are pulled back to the element's
 * parent cell for the binning and derivatives are with respect to the caller's coordinates. */
typedef struct fx_macro_element fx_macro_element;
int fx_macro_element_create(fx_ctx* ctx, int sd, int n, int variant, double scale,
                            const double* parent_verts, int ncell, const double* cell_verts,
                            int nmacro, const int* cell_node_map, const double* cell_scale,
                            int ndof, int vdim, const double* coeffs, fx_macro_element** elem);
int fx_macro_element_destroy(fx_macro_element* elem);
int fx_macro_element_set_coeffs(fx_macro_element* elem, int ndof, int vdim, const double* coeffs);
int fx_macro_tabulate_batch(fx_ctx* ctx, const fx_macro_element* elem, int order,
                            int64_t nreq, int npts, const double* pts,
                            const double* verts, double* out, void* stream);

/* ---- general tensor products and sub-entities --------------------------
 * TensorProductElement.tabulate (FIAT/tensor_product.py:231-336) for ANY two factors whose tables are on
 * the device -- simplex elements (fx_tabulate_batch), 1-D Lagrange (fx_line_tabulate_batch), products
 * themselves:  out[r][t][a*rowsB + b][c][p] = A[r][tA][a][cA][p] * B[r][tB][b][cB][p], alpha_t =
 * (alpha_A, alpha_B) over all |alpha| <= order in mis() order; at most one factor is vector-valued
 * (two: FX_ENOTIMPL, as the reference :271-272).  tabA device [nreq][C(sdA+order,sdA)][rowsA][vdimA][npts]
 * (all of the factor's tables up to `order`; sd 0 = a point factor with one table), tabB likewise,
 * out device [nreq][C(sdA+sdB+order, sdA+sdB)][rowsA*rowsB][max(vdimA,vdimB)][npts]. */
int fx_table_outer_batch(fx_ctx* ctx, int order, int sdA, int sdB, int64_t nreq, int npts, int rowsA,
                         int vdimA, int rowsB, int vdimB, const double* tabA, const double* tabB,
                         double* out, void* stream);
/* The affine map behind ``tabulate(order, points, entity=(dim, id))`` (FIAT/finite_element.py:181-197,
 * reference_element.py:570-609) for a whole batch: out[i] = M in[i] + b, M host [dout][din] (din = 0: every
 * point is the vertex b), in device [n][din], out device [n][dout]. */
int fx_map_points(fx_ctx* ctx, int din, int dout, const double* M, const double* b, int64_t n,
                  const double* in, double* out, void* stream);

/* ---- 1-D Jacobi polynomials ---------------------------------------------
 * FIAT/jacobi.py eval_jacobi_batch (:47-74) for order == 0 and eval_jacobi_deriv_batch (:85-102)
 * for order >= 1:  out[k][p] = d^order/dx^order P_k^{(a,b)}(xs[p]),  k = 0..n  (rows k < order are zero).
 * xs device [npts], out device [n+1][npts].  n <= 96. */
int fx_jacobi_batch(fx_ctx* ctx, double a, double b, int n, int order, int64_t npts, const double* xs,
                    double* out, void* stream);

/* ---- multi-GPU: reassembling the tables ----------------------------------
 * No reference counterpart (the reference is single-process NumPy): requests are independent, rank g of
 * N (one process per GPU, one fx_ctx each) tabulates a contiguous block, and only if the consumer wants all
 * tables on every GPU are the blocks exchanged -- RCCL over xGMI, bound at run time (no link dependency).
 *   fx_comm_available   FX_OK if RCCL could be loaded (call on every rank and agree BEFORE fx_comm_create:
 *                       communicator creation is collective and blocks until all ranks have joined);
 *   fx_comm_unique_id   one rank produces the 128-byte id and hands it to the others out of band
 *                       (torch.distributed store, MPI, a file);
 *   fx_comm_create      collective over the nranks processes.
 * fx_allgather_tables: rank p's block send[count] (device doubles) arrives at recv[p*stride + offset] on
 * every rank; recv holds nranks*stride doubles.  stride == count, offset == 0 is the plain all-gather of
 * equal blocks; a chunked, compute-overlapped gather passes the chunk's offset inside each rank's block
 * (offset + count <= stride).  send may alias the caller's own block of recv (in place).
 *   FX_GATHER_RING    ncclAllGather (blocks must tile recv);
 *   FX_GATHER_DIRECT  grouped send/recv with every peer: all 7 xGMI links of a GPU at once.
 * recv_count = doubles the caller owns at recv: (nranks-1)*stride + offset + count must not exceed it (FX_EINVAL).
 * Enqueued on `stream`; the exchange itself does not synchronise.  fx_comm_destroy waits for the streams exchanges
 * were enqueued on before it destroys the communicator (ABI version 2). */
typedef struct fx_comm fx_comm;
#define FX_COMM_ID_BYTES 128
#define FX_GATHER_RING 0
#define FX_GATHER_DIRECT 1
int fx_comm_available(void);
int fx_comm_unique_id(unsigned char* id /* host [FX_COMM_ID_BYTES] */);
int fx_comm_create(fx_ctx* ctx, int nranks, int rank, const unsigned char* id, fx_comm** comm);
int fx_comm_destroy(fx_comm* comm);
int fx_allgather_tables(fx_comm* comm, const double* send, double* recv, int64_t count, int64_t stride,
                        int64_t offset, int64_t recv_count, int algo, void* stream);

/* Synchronises `stream` and reports a scheduling failure of the dynamically scheduled kernels (their
 * work queue gives up after ~1 s instead of hanging the GPU; the affected launch's output is then
 * incomplete and this returns FX_EHIP).  The facade calls it wherever it copies tables to the host. */
int fx_ctx_check(fx_ctx* ctx, void* stream);

/* ---- measurement helpers ---------------------------------------------------------
 * Time `reps` launches of fx_tabulate_batch with HIP events on `stream`;
 * returns average milliseconds per launch in *ms. */
int fx_time_tabulate_batch(fx_ctx* ctx, const fx_element* elem, int order,
                           int64_t nreq, int npts, const double* pts,
                           const double* verts, double* out, void* stream,
                           int reps, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* FIAT_AMD_H */
