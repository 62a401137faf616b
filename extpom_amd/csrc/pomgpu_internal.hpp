// pomgpu_internal.hpp -- shared by the HIP kernel files and the C-ABI implementation.
//
// Data layout in HBM: the device mirrors ARE the reference's COMMON blocks (pom.h_dist:46-640):
// one allocation per block, arrays at slot*n2 / slot*n3, column-major with i contiguous and the
// reference's leading dimensions (im_local, jm_local).  Kernels map threadIdx.x -> i, so every
// global access of a wavefront is a contiguous 512-byte row segment.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdlib.h>
#include "pom_layout.h"

#ifndef COL_ROWS
#define COL_ROWS 4                                          /* rows per workgroup of the column kernels */
#endif
#ifndef COL_WX
#define COL_WX 1                                            /* wavefronts side by side along i per workgroup row */
#endif
#define POMGPU_NSCR3 8     // 3-D scratch arrays (the reference's automatic arrays that survive fusion)
#define POMGPU_NSCR2 8     // 2-D scratch arrays
#define POMGPU_NCOEF2 24   // derived 2-D coefficient arrays
#define POMGPU_MAXREC 8
#define POMGPU_KBMAX 128   // per-column private arrays in the tridiagonal kernels
#define POMGPU_CTX_2D 1
// bits of the device error word (pomgpu_ctx::d_err)
#define POMGPU_DERR_VELOCITY 1   // check_velocity: vamax > vmaxl (advance.f:631-637)
#define POMGPU_DERR_BARRIER 2    // k_ext_loop gave up at its grid barrier
#define POMGPU_NGEN 7        // arrays of the external mode that exist in two generations (enum pomgpu_x2)

// Kernel parameters, passed by value with every launch.
struct KP {
  int im, jm, kb, imm1, jmm1, kbm1, kbm2, iml, jml;
  int W, E, S, N;            // 1 where the tile edge is a physical boundary (n_west == -1 ...)
  size_t n2, n3;
  size_t a3;                 // distance between two arrays of blk3d in HBM, doubles (n3 + padding, pomgpu_create)
  double *b1, *b2, *b3, *bd; // device mirrors of blk1d, blk2d, blk3d, bdry
  double *r1;                // 1.0 / b1 element by element (k_coef_static): reciprocals of dz, dzz for divi()
  double *s3[POMGPU_NSCR3];
  double *s2[POMGPU_NSCR2];
  double *c2[POMGPU_NCOEF2]; // derived 2-D coefficient arrays (enum pomgpu_coef2), see k_tile.hip
  size_t bdoff[80];          // offset of every bdry member inside bd
  // blkcon scalars used on the hot path
  double alpha, dte, dti, dti2, dte2, grav, kappa, ramp, rfe, rfn, rfs, rfw, rhoref, sbias, small_,
         tbias, tprni, umol, horcon, ispi, isp2i, smoth, sw, time, vmaxl;
  int mode, ntp, nadv, nbct, nbcs, nitera, npg, isplit, iext, iint, iend;
  int g_rb, g_nbx, g_bpl;    // launch geometry of the banded cell kernels (set_band_geometry)
  int g_lin;                 // row-sharing column kernels: tiles whose block-rows leave the eight XCD bands under 90 % full take the balanced order (HALO_XCD_DECODE_R)
  int g_strip;               // row-sharing column kernels: an XCD walks its band in strips this many workgroups wide (0: row by row)
  // host-evaluated loop invariants (libm pow): solver.f:1273, :1297
  double const1_profq, cb_profq;
  // current (read) and next (written) generation of ua, va, d, el, elb, uab, vab for the external-mode kernels
  // (k_ext.hip); identical and equal to the blk2d arrays except inside the fused external step
  double *x2[POMGPU_NGEN], *y2[POMGPU_NGEN];
  // baropg_mcc's extra ghosts: rho4th(0,j,k) [kb x jml], rho4th(i,0,k) [kb x iml], d4th(0,j) [jml], d4th(i,0) [iml]
  double *g4[4];
  // the three 0/1 masks of a cell in one byte (bit 0 fsm, 1 dum, 2 dvm), k_coef_static: the fused external substep
  // reads 2 bytes instead of 4 doubles per cell; (double)bit * x is the reference's mask multiply, bit for bit
  unsigned char *m8;
};
enum pomgpu_x2 { X2_ua, X2_va, X2_d, X2_el, X2_elb, X2_uab, X2_vab };

// ---- storage type of the 3-D arrays -----------------------------------------------------------------------------------
// The product (libpomgpu.so) keeps every array in fp64, like the reference.  -DPOMGPU_STORE_F32 builds the variant of
// BASELINE configs[4] ("fp32 internal mode / fp64 external mode"): the arrays of blk3d and the 3-D scratch arrays are
// STORED as fp32 (half the bytes of the HBM-bound internal mode), every value is widened to fp64 on load and rounded on
// store, all arithmetic -- the Thomas solves, the vertical integrals, the whole 2-D external mode with its arrays -- stays
// fp64.  Same sources: the accessors below hand out proxies instead of references, the buffer helpers move 4 bytes per
// element, upload / download convert.  Array k of blk3d still starts k * a3 * 8 bytes into the block (it fills the first
// half of its slot), so every `double *` that names a 3-D array keeps its meaning as a handle.  Not in this variant:
// tiles / transports, the output writer, the two-columns-per-lane aam kernel (they index 3-D arrays as raw doubles).
#ifdef POMGPU_STORE_F32
typedef float pomgpu_st;
#else
typedef double pomgpu_st;
#endif
#define POMGPU_ST_BYTES ((unsigned)sizeof(pomgpu_st))
#ifdef POMGPU_STORE_F32
struct Ref3 {
  pomgpu_st *p;
  __device__ __forceinline__ operator double() const { return (double)*p; }
  __device__ __forceinline__ const Ref3 &operator=(double v) const { *p = (pomgpu_st)v; return *this; }
  __device__ __forceinline__ const Ref3 &operator=(const Ref3 &o) const { *p = *o.p; return *this; }
};
#define REF3(ptr, idx) (Ref3{(pomgpu_st *)(ptr) + (idx)})
#else
#define REF3(ptr, idx) ((ptr)[(idx)])
#endif

// ---- Fortran-style accessors (1-based), `P` is the KP in scope ------------------------------
#define IX2(i, j) ((size_t)((j)-1) * (size_t)P.iml + (size_t)((i)-1))
#define IX3(i, j, k) ((size_t)((k)-1) * P.n2 + IX2(i, j))
// blk1d (z, zz, dz, dzz) is indexed by the level only: the address is uniform across the wavefront.  Read through the
// CONSTANT address space the loads become scalar loads (s_load, counted by lgkmcnt).  As plain global loads the compiler
// cannot prove the arrays are not written by the kernel's own stores and issues VECTOR loads of a uniform address -- which
// queue behind the prefetch batch of the next level and, vmcnt being in-order, make the first use of dz(k) wait for that
// whole batch: the software pipeline of every column kernel was serialised by it (ISA of k_advt2_col: s_waitcnt vmcnt(0)
// in the middle of each iteration).  The arrays never change while a kernel runs (upload writes them).
#if !defined(POMGPU_EMU) && !defined(POMGPU_F1_GLOBAL)
typedef const double __attribute__((address_space(4))) *pomgpu_cptr;
#define F1(name, k) ((pomgpu_cptr)(size_t)P.b1)[(size_t)P1_##name * P.kb + ((k)-1)]
#define R1(name, k) ((pomgpu_cptr)(size_t)P.r1)[(size_t)P1_##name * P.kb + ((k)-1)]
#else
#define F1(name, k) P.b1[(size_t)P1_##name * P.kb + ((k)-1)]
#define R1(name, k) P.r1[(size_t)P1_##name * P.kb + ((k)-1)]
#endif
#define F2(name, i, j) P.b2[(size_t)P2_##name * P.n2 + IX2(i, j)]
#define F3(name, i, j, k) REF3(P.b3 + (size_t)P3_##name * P.a3, IX3(i, j, k))
#define A2(name) (P.b2 + (size_t)P2_##name * P.n2)
#define A3(name) (P.b3 + (size_t)P3_##name * P.a3)
#define G3(p, i, j, k) REF3(p, IX3(i, j, k))
#define G2(p, i, j) (p)[IX2(i, j)]

enum pom_bdry_slot_dev {
#define POMGPU_BD_(name, shape) PB_##name,
  POM_BDRY(POMGPU_BD_)
#undef POMGPU_BD_
  PB__count
};
#define BD1(name, a) P.bd[P.bdoff[PB_##name] + (size_t)((a)-1)]
#define BDJ(name, a, k) P.bd[P.bdoff[PB_##name] + (size_t)((k)-1) * P.jml + (size_t)((a)-1)]
#define BDI(name, a, k) P.bd[P.bdoff[PB_##name] + (size_t)((k)-1) * P.iml + (size_t)((a)-1)]

__device__ __forceinline__ double sq(double x) { return x * x; }

// Derived 2-D coefficients: sums/products of grid metrics that every face-flux formula of the
// reference evaluates as a unit (e.g. "(dy(i,j)+dy(i-1,j))", "0.25*(dy+dy)*(dt+dt)"), formed once
// instead of once per level and per face.  *X live on the west face of cell (i,j), *Y on its south face.
enum pomgpu_coef2 {
  C2_HSX, C2_HSY,      // h(i,j)+h(i-1,j), h(i,j)+h(i,j-1)                    (static)
  C2_DYSX, C2_DXSX,    // dy(i,j)+dy(i-1,j), dx(i,j)+dx(i-1,j)                (static)
  C2_DXSY, C2_DYSY,    // dx(i,j)+dx(i,j-1), dy(i,j)+dy(i,j-1)                (static)
  C2_CMX, C2_CMY,      // 0.25*DYSX*(dt(i,j)+dt(i-1,j)), 0.25*DXSY*(dt(i,j)+dt(i,j-1))   (follow dt)
  C2_HEA, C2_HFA,      // (h+etb)*art, (h+etf)*art                            (follow etb, etf)
  C2_DTSX, C2_DTSY,    // dt(i,j)+dt(i-1,j), dt(i,j)+dt(i,j-1)                (follow dt)
  C2_DT4,              // dt(i,j)+dt(i-1,j)+dt(i,j-1)+dt(i-1,j-1)             (follow dt)
  C2_DX4, C2_DY4,      // the same four-point sums of dx and dy               (static)
  C2_CVA, C2_CVB,      // dy(i+1,j)-dy(i-1,j), dx(i,j+1)-dx(i,j-1)            (static; curvature terms)
  C2_R2DXSX, C2_R2DYSY,// 2.0/DXSX, 2.0/DYSY                                  (static; realvertvl's dxl/dyb)
  C2_RDX, C2_RDY,      // RN(1/dx), RN(1/dy): the reciprocals divi() needs         (static; k_aam_pair's four divisions per cell)
  C2__count
};
static_assert(C2__count <= POMGPU_NCOEF2, "raise POMGPU_NCOEF2");
#define K2(name, i, j) P.c2[C2_##name][IX2(i, j)]

// Neighbour-lane access for stencils along i.  A wavefront owns 64 consecutive i of one row; the
// value its western / eastern neighbour lane holds replaces a second global load of the same word.
// The two edge lanes of the wavefront have no such neighbour and evaluate `fb` (a load or a
// recomputation) instead.  Every lane of the wavefront must reach the call.
// Whole-wavefront shift by one lane as two DPP moves (wave_shr:1 / wave_shl:1, GFX9 encodings 0x138 /
// 0x130): a plain VALU operation, no trip through the LDS crossbar as with __shfl_up/__shfl_down
// (ds_bpermute), so no lgkmcnt wait and no batch of live results.  Edge lanes keep their own value,
// exactly as the shuffles do (tools/micro/dpp_shift.hip checks both on the device).
#ifndef POMGPU_EMU
__device__ __forceinline__ double wave_up1(double x) {      // lane n <- lane n-1
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_dn1(double x) {      // lane n <- lane n+1
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
#else
static inline double wave_up1(double x) { return x; }
static inline double wave_dn1(double x) { return x; }
#endif
template <class F> __device__ __forceinline__ double lane_w(double x, F fb) {
  const double t = wave_up1(x);
  return (threadIdx.x == 0) ? fb() : t;
}
template <class F> __device__ __forceinline__ double lane_e(double x, F fb) {
  const double t = wave_dn1(x);
  return (threadIdx.x == blockDim.x - 1) ? fb() : t;
}

// Halo-lane variant: a wavefront covers 64 consecutive columns, lanes 1..62 own an output column,
// lanes 0 and 63 only load their column and feed the shuffles (wave w covers columns 62w..62w+63,
// HALO_COL below).  No lane ever takes a fallback path, so there are no dependent second-phase
// loads and no flux is evaluated twice by a whole wavefront for the sake of one lane.
#ifdef POMGPU_EMU   // host emulation runs one lane at a time: neighbour values are recomputed from memory
template <class F> __device__ __forceinline__ double halo_w(double, F fb) { return fb(); }
template <class F> __device__ __forceinline__ double halo_e(double, F fb) { return fb(); }
#else
template <class F> __device__ __forceinline__ double halo_w(double x, F) { return wave_up1(x); }
template <class F> __device__ __forceinline__ double halo_e(double x, F) { return wave_dn1(x); }
#endif
// nothing may be scheduled across this point (keeps a block of prefetch loads together and early)
#ifdef POMGPU_EMU
#define SCHED_FENCE()
#else
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// a value that is the same in every lane of the wavefront (e.g. threadIdx.y with 64-wide rows), told to the compiler
#ifndef POMGPU_EMU
#define WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define WAVE_UNIFORM(x) (x)
#endif
#define HALO_LANE (int)((blockIdx.x * blockDim.x + threadIdx.x) & 63)

// ---- buffer addressing for the column kernels ---------------------------------------------------------------------
// A column kernel reads the same cell of ~10 arrays at every level.  With plain pointers the compiler keeps one
// 64-bit address per (array, row) in VGPRs and advances each of them every level (PMC / ISA of k_advt2_col<2>: 44
// v_lshl_add_u64 per level out of 445 instructions, ~40 VGPRs of addresses).  Buffer loads split the address the way
// the data is laid out: a resource descriptor per array (4 SGPRs, constant), a per-lane byte offset of the cell inside
// a level (one VGPR per row used, constant for the whole column) and the level's byte offset as the scalar offset
// (one SGPR shared by ALL arrays, advanced by the scalar unit).  No vector instruction is spent on addresses.
// Offsets are 32 bits: a 3-D array must stay below 4 GiB (checked in pomgpu_create).
#ifndef POMGPU_EMU
typedef unsigned int pomgpu_u32x2 __attribute__((ext_vector_type(2)));
struct BufA { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ BufA buf_of(const double *p, size_t doubles) {
  BufA b;
  b.r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (unsigned)(doubles * POMGPU_ST_BYTES), 0x00020000);   // raw buffer, no swizzle
  return b;
}
#ifndef POMGPU_STORE_F32
__device__ __forceinline__ double bld(const BufA &b, unsigned voff, unsigned soff) {
  const pomgpu_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(b.r, (int)voff, (int)soff, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void bst(const BufA &b, unsigned voff, unsigned soff, double x) {
  pomgpu_u32x2 v;
  v.x = (unsigned)__double2loint(x); v.y = (unsigned)__double2hiint(x);
  __builtin_amdgcn_raw_buffer_store_b64(v, b.r, (int)voff, (int)soff, 0);
}
#else
__device__ __forceinline__ double bld(const BufA &b, unsigned voff, unsigned soff) {
  return (double)__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b.r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bst(const BufA &b, unsigned voff, unsigned soff, double x) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)x), b.r, (int)voff, (int)soff, 0);
}
#endif
#else
struct BufA { double *p; };
static inline BufA buf_of(const double *p, size_t) { BufA b; b.p = (double *)p; return b; }
static inline double bld(const BufA &b, unsigned voff, unsigned soff) { return voff >= 0xFFFFFFF0u ? 0. : b.p[((size_t)voff + soff) >> 3]; }
static inline void bst(const BufA &b, unsigned voff, unsigned soff, double x) { if (voff < 0xFFFFFFF0u) b.p[((size_t)voff + soff) >> 3] = x; }
#endif
// 2-D arrays are fp64 in every build: buffer helpers that do not follow the storage type of the 3-D arrays
#ifndef POMGPU_EMU
__device__ __forceinline__ BufA buf2_of(const double *p, size_t doubles) {
  BufA b;
  b.r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (unsigned)(doubles * 8), 0x00020000);
  return b;
}
__device__ __forceinline__ double bld2(const BufA &b, unsigned voff, unsigned soff) {
  const pomgpu_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(b.r, (int)voff, (int)soff, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void bst2(const BufA &b, unsigned voff, unsigned soff, double x) {
  pomgpu_u32x2 v;
  v.x = (unsigned)__double2loint(x); v.y = (unsigned)__double2hiint(x);
  __builtin_amdgcn_raw_buffer_store_b64(v, b.r, (int)voff, (int)soff, 0);
}
#else
static inline BufA buf2_of(const double *p, size_t) { BufA b; b.p = (double *)p; return b; }
static inline double bld2(const BufA &b, unsigned voff, unsigned soff) { return voff >= 0xFFFFFFF0u ? 0. : b.p[((size_t)voff + soff) >> 3]; }
static inline void bst2(const BufA &b, unsigned voff, unsigned soff, double x) { if (voff < 0xFFFFFFF0u) b.p[((size_t)voff + soff) >> 3] = x; }
#endif
// per-lane offset of a lane that must not store: outside every descriptor, the hardware drops the access (loads return 0;
// tools/micro/buf_oob.hip).  A branch around a store or load inside a software-pipelined loop costs far more than the
// wasted lanes: where the paths join the compiler no longer knows how many memory operations are outstanding and waits
// for vmcnt(0), i.e. for the prefetch batch that was just issued.
#define BOFF_NONE 0xFFFFFFF0u
#define BUF3(p) buf_of((p), P.n3)
#define BOFF2(i, j) (unsigned)(((unsigned)((j)-1) * (unsigned)P.iml + (unsigned)((i)-1)) * POMGPU_ST_BYTES)   /* byte offset of (i,j) inside a level */
#define LVB (unsigned)(P.n2 * POMGPU_ST_BYTES)                                              /* byte distance between two levels */

// ---- rows of a workgroup shared through LDS ---------------------------------------------------------------------------
// A column kernel with a j-stencil needs every such operand of rows j-1, j, j+1.  Loaded per wavefront (one row each)
// that is three wavefront loads per operand and level, and the counters say that is what bounds these kernels: the L1
// (TCP) spends ~21 accesses on every 8-byte-per-lane wavefront load and saturates at ~0.6 accesses per CU-cycle
// (k_advt2_col: 0.59, whatever the occupancy; with the neighbour-row loads taken out the kernel ran 26 % faster), and
// the re-requested rows are L2 hits only 40 % of the time (+45 % HBM-side reads).  Here the LDS_ROWS wavefronts of a
// workgroup load their OWN row of each shared operand, park it in an LDS slab and read the rows next to theirs from
// there; the two rows outside the workgroup (south of its first row, north of its last) are split among its wavefronts
// (ROWSHARE_SLOTS loads each).  One s_barrier per level; two slabs, so a wavefront may fill the next level's slab while
// its neighbours still read this level's.  Loads per wavefront and level: NS + NO + slots instead of 3*NS + NO.
// Slab rows: 0 = south halo, 1..LDS_ROWS = the workgroup's rows, LDS_ROWS+1 = north halo, LDS_ROWS+2 = sink for the
// slots a wavefront has no job for.  Host emulation (one lane at a time, no concurrency) reads the rows from memory.
// keeps the wavefronts of a workgroup on the same level of a level loop (no memory ordering implied or needed)
#ifdef POMGPU_EMU
#define PACE_BARRIER() ((void)0)
#else
#define PACE_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifndef LDS_ROWS
#define LDS_ROWS 8                                          /* rows per workgroup of the row-sharing kernels (kbench: 8 beats 4 and 6 by 8-14 %) */
#endif
#define ROWSHARE_SLOTS(NS) ((2 * (NS) + LDS_ROWS - 1) / LDS_ROWS)
#define ROWSHARE_ROWS (LDS_ROWS + 3)
// ROWS: the rows of the workgroup (LDS_ROWS everywhere but in k_advct_col's 4-row shape for low tiles)
template <int NS, int ROWS = LDS_ROWS> struct RowShare {
  static constexpr int SLOTS = (2 * NS + ROWS - 1) / ROWS;
  int r;                                   // this wavefront's row inside the workgroup (scalar)
  int ss, sn;                              // slab rows holding the southern / northern neighbour row of this wavefront's row
  int hop[SLOTS];                          // halo jobs of this wavefront: which shared operand ...
  int hrow[SLOTS];                         // ... into which slab row ...
  unsigned hoff[SLOTS];                    // ... from which cell (per-lane byte offset inside a level; BOFF_NONE = no job)
};
// j: the wavefront's row (may lie beyond jml in the last workgroup: such rows shadow row jml and store nothing),
// jc = min(j, jml), j0w: first row of the workgroup, i: the lane's (clamped) column
template <int NS, int ROWS = LDS_ROWS> __device__ __forceinline__ RowShare<NS, ROWS> rowshare_setup(const KP &P, int r, int j, int j0w, int i) {
  RowShare<NS, ROWS> S;
  S.r = r;
  S.ss = (j > 1 && j <= P.jml) ? r : r + 1;
  S.sn = (j < P.jml) ? r + 2 : r + 1;
  const int jsouth = j0w > 1 ? j0w - 1 : 1, jnorth = j0w + ROWS <= P.jml ? j0w + ROWS : P.jml;
#pragma unroll
  for (int q = 0; q < RowShare<NS, ROWS>::SLOTS; q++) {
    const int job = q * ROWS + r;
    const bool valid = job < 2 * NS;
    const int side = valid ? job / NS : 0;
    S.hop[q] = valid ? job % NS : 0;
    S.hrow[q] = valid ? (side ? ROWS + 1 : 0) : ROWS + 2;
    S.hoff[q] = valid ? BOFF2(i, side ? jnorth : jsouth) : BOFF_NONE;
  }
  return S;
}
// the array a halo job reads: chosen by a wavefront-uniform index; the pointer is pinned to scalar registers so that the
// descriptor built from it is one (a descriptor in vector registers costs a waterfall loop around every load)
template <int NS> __device__ __forceinline__ const double *rowshare_pick(const double *const (&b)[NS], int op) {
  const double *x = b[0];
#pragma unroll
  for (int o = 1; o < NS; o++)
    if (op == o) x = b[o];
#ifndef POMGPU_EMU
  const unsigned long long v = (unsigned long long)x;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  x = (const double *)(((unsigned long long)hi << 32) | lo);
#endif
  return x;
}

// ---- division by a divisor that does not change along the column ---------------------------------------------------
// a / b is the reference's operation (IEEE, correctly rounded); the GPU's macro for it is ~10 instructions, several of
// them slow (v_div_scale x2, v_rcp_f64, 5 fma, v_div_fmas, v_div_fixup).  The column kernels divide by the same
// metric sums at every level.  With y = RN(1/b) formed once (by a real division) the correctly rounded quotient is
//   q0 = RN(a*y); r0 = a - b*q0 (exact in one fma); q1 = RN(q0 + r0*y)   -- faithful (within one ulp of a/b)
//   r1 = a - b*q1 (exact);                        q  = RN(q1 + r1*y)   -- = RN(a/b): Markstein's theorem (y correctly
// rounded, q1 faithful; Muller et al., Handbook of Floating-Point Arithmetic, Thm. 4.9) -- five full-rate
// instructions, no special cases as long as nothing over- or underflows (metric sums and depths: never).
// tools/micro/f64_rates.hip compares it with the hardware quotient on 3e9 operand pairs; tests compare whole steps.
struct InvD { double b, y; };
__device__ __forceinline__ InvD inv_of(double b) { InvD d; d.b = b; d.y = 1.0 / b; return d; }
#ifndef POMGPU_EMU
// A zero numerator keeps its sign: masked flux numerators such as -am * ... * msk are -0.0 on land, a/b is then -0.0 (b > 0),
// but the corrections give (+0) + (-0) = +0 -- the product a*y has the quotient's sign, one select hands it through (array_equal
// cannot see the sign of a zero, a digest of the bytes can).  Not covered: non-zero numerators below 2^-969 in magnitude,
// whose residual a - b*q would be subnormal and no longer exact (the last bit may differ from a/b there) -- the numerators
// here are products of metrics, depths and O(1e-12 ... 1e3) field differences, eight hundred binary orders of magnitude away.
__device__ __forceinline__ double divi(double a, const InvD &d) {
  const double q0 = a * d.y;
  double r = __builtin_fma(-d.b, q0, a);
  double q = __builtin_fma(r, d.y, q0);
  r = __builtin_fma(-d.b, q, a);
  q = __builtin_fma(r, d.y, q);
  return a == 0. ? q0 : q;
}
#else
static inline double divi(double a, const InvD &d) { return a / d.b; }
#endif
// XCD-aware placement of the column kernels' workgroups (64 x 4 columns each).  Workgroups are dealt
// to the 8 XCDs round-robin in linear-id order, and every XCD has its own 4 MiB L2.  A plain
// (bx, by) grid therefore puts vertically adjacent workgroups on different XCDs and each re-fetches
// its two halo rows from HBM (6 rows read for 4 computed: PMC showed 10.5 array passes for 6-7
// algorithmic).  Here the launch is 1-D: XCD x owns the band of block-rows [x*rpx, (x+1)*rpx) and walks
// it row by row, so the ~3 block-rows an XCD has resident at any time are neighbours and the halo rows
// are L2 hits.  The mapping is only a performance hint: any dispatch order gives the same results.
// HALO_XCD_DECODE defines i0 (1-based column, halo-lane numbering) and j (1-based row), or returns.
// Strip order (P.g_strip > 0): the workgroups an XCD has resident at one time are a patch g_strip wide and several block-rows
// tall instead of 2-3 whole block-rows, so that the level a workgroup is on is also being read by the workgroups above /
// below it and beside it at about the same time.  Same-context A/B at 2048 columns (34 workgroups per block-row;
// profiles/round2_strip_order.txt): width 2 +6..10 %, 4 +-0, 8 -2..5 %, 11-12 -2.5..5 % (advct, advq), 16 -1..3 % against
// whole rows; set_band_geometry picks the width nearest 12 that divides the row evenly.
#define HALO_XCD_ORDER                                                                    \
  if (P.g_strip > 0) {                                                                    \
    const int st__ = P.g_strip, nfull__ = nbx__ / st__, wlast__ = nbx__ - nfull__ * st__; \
    int s__ = m__ / (rpx__ * st__);                                                       \
    if (s__ > nfull__) s__ = nfull__;                                                     \
    const int rem__ = m__ - s__ * rpx__ * st__, w__ = s__ < nfull__ ? st__ : (wlast__ > 0 ? wlast__ : 1); \
    byl__ = rem__ / w__; bxg__ = s__ * st__ + rem__ % w__;                                \
    if (bxg__ >= nbx__ || byl__ >= rpx__) return;                                         \
  } else {                                                                                \
    byl__ = m__ / nbx__; bxg__ = m__ % nbx__;                                             \
    if (byl__ >= rpx__) return;                                                           \
  }
#define HALO_XCD_DECODE HALO_XCD_DECODE_R(COL_ROWS)
// Tiles whose block-rows fill less than 90 % of the eight bands (a 194-row tile of a 1 x 8 split: 25 block-rows of 8 -> bands of
// 4, 4, ..., 1, 0 block-rows: one XCD idle, one at a quarter) take the BALANCED order instead (P.g_lin): the workgroups in
// strip-major order (strip, block-row, column inside the strip -- a strip is g_strip wide, or the whole row), XCD x owns the
// contiguous share [x*T/8, (x+1)*T/8) of that sequence: shares differ by one workgroup at most, an XCD's workgroups are still
// neighbours.  tools/lin_ab.sh, one tile of 2048x1536x50, the four kernels together: 1 x 8 split (25 block-rows) 1.83 -> 1.67 ms,
// 1 x 4 (49) 3.11 -> 3.03, 1 x 2 (97 block-rows, 93 % full) 5.47 -> 5.65: the banded order stays where the bands are nearly full.
#define HALO_XCD_DECODE_R(ROWS__)                                                         \
  const int g__ = (int)(blockIdx.x * blockDim.x + threadIdx.x);                           \
  const int L__ = g__ >> 6, nwx__ = (P.iml + 61) / 62, nbx__ = (nwx__ + COL_WX - 1) / COL_WX; \
  const int nby__ = (P.jml + ROWS__ - 1) / ROWS__;                                        \
  const int rpx__ = (nby__ + 7) / 8, m__ = L__ >> 3;                                      \
  int by__, bxg__;                                                                        \
  if (P.g_lin && 10 * nby__ < 72 * rpx__) {                                               \
    const int x__ = L__ & 7, T__ = nby__ * nbx__;                                         \
    const int c0__ = (int)(((long)x__ * T__) >> 3), c1__ = (int)(((long)(x__ + 1) * T__) >> 3); \
    const int t__ = c0__ + m__;                                                           \
    if (t__ >= c1__) return;                                                              \
    const int st__ = P.g_strip > 0 ? P.g_strip : nbx__, per__ = st__ * nby__, s__ = t__ / per__; \
    const int rem__ = t__ - s__ * per__, w__ = (s__ + 1) * st__ <= nbx__ ? st__ : nbx__ - s__ * st__; \
    by__ = rem__ / w__; bxg__ = s__ * st__ + rem__ % w__;                                 \
  } else {                                                                                \
    int byl__;                                                                            \
    HALO_XCD_ORDER                                                                        \
    by__ = (L__ & 7) * rpx__ + byl__;                                                     \
    if (by__ >= nby__) return;                                                            \
  }                                                                                       \
  const int lane = g__ & 63;                                                              \
  const int i0 = (bxg__ * COL_WX + (int)threadIdx.y % COL_WX) * 62 + lane;                \
  const int j = by__ * ROWS__ + (int)threadIdx.y / COL_WX + 1;
static inline dim3 grid1_halo_r(const KP &P, int rows) {
  const int nwx = (P.iml + 61) / 62, nbx = (nwx + COL_WX - 1) / COL_WX, nby = (P.jml + rows - 1) / rows, rpx = (nby + 7) / 8;
  if (P.g_lin && 10 * nby < 72 * rpx) return dim3((unsigned)(8 * ((nby * nbx + 7) / 8)), 1, 1);
  return dim3((unsigned)(8 * rpx * nbx), 1, 1);
}
static inline dim3 grid1_halo(const KP &P) { return grid1_halo_r(P, COL_ROWS); }
static inline dim3 blk_col_r(int rows) { return dim3(64, rows * COL_WX, 1); }
static inline dim3 blk_col() { return dim3(64, COL_ROWS * COL_WX, 1); }
#define HALO_COL (int)(((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 62 + ((blockIdx.x * blockDim.x + threadIdx.x) & 63))
static inline dim3 grid2_halo(const KP &P) { return dim3((P.iml + 61) / 62, (P.jml + 3) / 4, 1); }

// thread -> (i,j[,k]) maps (1-based); blockDim.x runs along i
#define TID_I (int)(blockIdx.x * blockDim.x + threadIdx.x + 1)
#define TID_J (int)(blockIdx.y * blockDim.y + threadIdx.y + 1)
#define TID_K (int)(blockIdx.z * blockDim.z + threadIdx.z + 1)

// ---- host side ---------------------------------------------------------------------------------
// Developer switches: environment variables POMGPU_<NAME>, read ONCE when a context is created
// (pomgpu_create) into the context -- no getenv on the launch path, and every context of a process keeps the switch set
// it was created under.  pomgpu_debug_switch (include/pomgpu.h) changes one of them afterwards: tools/kbench.py lets
// variants take turns on ONE context with it, tests give the ranks of one process different switch sets.
#define POMGPU_SWITCHES(X)                                                                                                    \
  X(THOMAS_SCRATCH) X(NO_PAIR) X(EXT_SPLIT) X(ADVAVE_SEPARATE) X(EXT_RIM_KERNEL) X(EXT_LOOP) X(EXT_NOMARCH) X(EXT_MARCH)      \
  X(EXT_ROWS) X(EXT_NOPAIR) X(EXT_PAIR) X(EXT_ROWS2) X(EXT_TWO_SETS) X(EXT_AREAS_LOAD) X(BAROPG_CELLS) X(VERTVL_CELLS)        \
  X(NO_LIN) X(ADVQ_SINGLE) X(ADVT2_SINGLE) X(REALVERTVL_CELLS) X(RHO_ROUNDTRIP) X(TAU_ARRAYS) X(PROFQ_ROWS8) X(PROFQ_ROWS2)    \
  X(PROFQ_NOPACE) X(COL_STRIP) X(BAND_BYTES) X(PAD3) X(IO_SYNC) X(ADVCT_SPLIT) X(ADVQ_EXCHANGE) X(PROD_FULL) X(QFILTER_SPLIT) \
  X(UV_FULL_EXCHANGE) X(NO_OVERLAP) X(NO_SIDE_COMM) X(WR_MAIN) X(WIDE_W) X(WIDE_FULL) X(DEBUG_ALLOC) X(TEST_SPLIT_FAIL_RANK)  \
  X(EDGE_SPLIT) X(WR_NODEFER) X(EXT_RING_FIRST) X(TUNE_FORCE) X(NO_TWIN) X(RIM_MAIN) X(RIM_RESULTS_MAIN) X(SUM2D_OFF) X(ADVCT_ROWS4) X(ADVCT_ROWS8)
enum pomgpu_sw {
#define POMGPU_SW_(name) SW_##name,
  POMGPU_SWITCHES(POMGPU_SW_)
#undef POMGPU_SW_
  SW__count
};
struct pomgpu_switches {
  unsigned char on[SW__count];   // the variable is set (whatever its value: presence is the switch)
  long val[SW__count];           // atol of its value (switches that carry a number: EXT_ROWS, COL_STRIP, WIDE_W ...)
};
static const char *const POMGPU_SW_NAMES[SW__count] = {
#define POMGPU_SW_(name) #name,
  POMGPU_SWITCHES(POMGPU_SW_)
#undef POMGPU_SW_
};
static inline void pomgpu_switches_read(pomgpu_switches &s) {
  for (int n = 0; n < SW__count; n++) {
    char var[64] = "POMGPU_";
    size_t k = 7;
    for (const char *q = POMGPU_SW_NAMES[n]; *q && k + 1 < sizeof var; q++) var[k++] = *q;
    var[k] = 0;
    const char *e = getenv(var);
    s.on[n] = e ? 1 : 0;
    s.val[n] = e ? atol(e) : 0;
  }
}
// order-independent digest of the switches that every rank of a decomposition must share (they choose which message
// rounds exist and on which stream / communicator they run): the ranks compare it before any collective depends on it
static inline unsigned pomgpu_switches_collective_digest(const pomgpu_switches &s) {
  static const int coll[] = {SW_ADVCT_SPLIT, SW_ADVQ_EXCHANGE, SW_PROD_FULL, SW_QFILTER_SPLIT, SW_UV_FULL_EXCHANGE, SW_NO_OVERLAP,
                             SW_NO_SIDE_COMM, SW_WR_MAIN, SW_WIDE_W, SW_WIDE_FULL, SW_EXT_SPLIT, SW_ADVAVE_SEPARATE, SW_EDGE_SPLIT, SW_WR_NODEFER, SW_RIM_MAIN, SW_ADVT2_SINGLE, SW_RIM_RESULTS_MAIN, SW_SUM2D_OFF};
  unsigned h = 2166136261u;
  for (size_t n = 0; n < sizeof coll / sizeof coll[0]; n++) {
    h = (h ^ (unsigned)(s.on[coll[n]] ? 1 + coll[n] : 0)) * 16777619u;
    h = (h ^ (unsigned)s.val[coll[n]]) * 16777619u;
  }
  return h & 0x3fffffffu;
}
#define SW(c, NAME) ((c)->sw.on[SW_##NAME] != 0)
#define SWV(c, NAME) ((c)->sw.val[SW_##NAME])
struct ProfEntry { const char *name; long launches; double ms; };
struct pomgpu_ctx;

// Direction order of everything that talks to up to eight neighbours: W E S N SW SE NW NE.  What a tile sends
// towards direction d arrives at the neighbour as coming from POMGPU_OPP[d].
static const int POMGPU_OPP[8] = {1, 0, 3, 2, 7, 6, 5, 4};
// The library's own transport (pomgpu_set_transport / pomgpu_rccl_init): staging buffers per direction and the
// mover -- a host callback (tests) or grouped ncclSend/ncclRecv on the kernels' stream (transport.hip).
struct pomgpu_transport {
  int on;
  int nbr[8];                // neighbour ranks, -1 = none
  void (*fn)(void *, const double *const *, const size_t *, double *const *, const size_t *);   // pomgpu_transport_fn
  void *user;
  void *rccl;                // RcclComm* (transport.hip), NULL with a callback transport
  double *send[8], *recv[8];
  size_t cap[8];             // capacity of each staging buffer, doubles
  double *send2[8], *recv2[8];   // staging buffers of the side stream's rounds (they overlap rounds on the main stream)
  size_t cap2[8];
  int fn_ordered;            // the callback mover enqueues on pomgpu_current_stream() itself (pomgpu_transport_stream_ordered)
  int side_agreed;           // EVERY rank of the decomposition can serve rounds on the side stream (collective decision, transport.hip)
  int wr_side;               // ... and none of them asked for the wr exchange on the main stream (POMGPU_WR_MAIN)
  long rounds_side;          // message rounds served on the side stream
  long rounds;               // message rounds served so far
};
// One rectangular block copy: ni x nj doubles from src (row stride ld_s) to dst (row stride ld_d).
struct RectJob { const double *src; double *dst; int ld_s, ld_d, ni, nj; };
// The wide-halo external mode (pomgpu_set_wide_external): a 2-D-only context `x` of the tile extended by `w`
// cells towards every neighbour; index i of the tile is index i + ox of the extended tile.
struct RectGroup { int first, count, mni, mnj; };   // jobs of similar extent share one launch
struct RectTable { RectJob *dev; int ngroups; RectGroup g[8]; };
struct pomgpu_wide {
  int on, w, ox, oy;
  pomgpu_ctx *x;
  RectTable gather_pack, gather_unpack, scatter;   // per internal step
  size_t scount[8], rcount[8];
  // the same gather in two parts: EARLY = every array that is final when the step starts (most of the bytes; side stream,
  // beside lateral_viscosity), LATE = the vertical integrals lateral_viscosity / mode_interaction produce (main stream)
  RectTable early_pack, early_unpack, late_pack, late_unpack;
  size_t e_scount[8], e_rcount[8], l_scount[8], l_rcount[8];
  int split;                 // the two-part tables exist and the transport can serve the side stream
  int pending;               // the 2-D state of the running external loop lives in x (between wide_begin and the last substep)
  int static_done;           // every blk2d array (grid metrics, masks ...) has been widened since the last upload
};

struct pomgpu_ctx {
  KP P;
  pomgpu_switches sw;        // developer switches, read from the environment once at pomgpu_create (never on the launch path)
  int device;
  hipStream_t stream;
  bool own_stream;
  // Second stream (pomgpu_set_wide_external creates it): message rounds whose arrays are ready early, or whose ghost cells
  // nobody waits for, run there beside the kernels of the main stream -- their pack / unpack kernels included -- over a
  // second communicator, joined by events (no host synchronisation).  `cur` is the stream LAUNCH and the profiler use:
  // the main stream except while work for the side stream is being enqueued.
  hipStream_t cur, side;
  hipEvent_t ev_fork, ev_early, ev_side;
  // Message rounds of exchange points that run on the side stream beside the main stream's kernels (pomgpu_api.hip "rim rounds"):
  // ev_r1 = advct's edge lines have arrived, ev_r2 = advx, advy, aam have their ghost cells (r2_pending: the main stream has not
  // waited yet), ev_r8 = the velocity rounds that end mode_internal (uf, vf, wubot, wvbot; then ub, vb and level kb of u, uf, v, vf)
  // ev_rw = w has its ghost cells, ev_rq = the turbulence arrays (q2, q2b, q2l, q2lb, km, kh, kq, l, dtef), ev_rts = t, tb, s, sb, rho
  hipEvent_t ev_r1, ev_r2, ev_r8, ev_rw, ev_rq, ev_rts;
  int r2_pending, r8_pending, rw_pending, rq_pending, rts_pending;
  int early_started;         // this step's early part of the wide exchange is in flight on the side stream (ev_early ends it)
  double *tune_block;        // pomgpu_tune_placement: blk3d and the 3-D scratch arrays live in ONE allocation with room in front ...
  size_t tune_front;         // ... and start this many doubles into it
  int tune_kmax;             // the room in front: this many units of two arrays
  size_t tune_a0, tune_amax; // the distance between the arrays of blk3d as it was, and the wider one the allocation has room for
  int dev3_handed;           // pomgpu_device_3d has given out an address of a 3-D array: pomgpu_tune_placement, which moves them, refuses from then on
  int broken;                // a relayout of the 3-D arrays failed half way (pomgpu_tune_placement): the mirrors no longer hold the state, every hot-path entry point refuses
  int wr_deferred;           // pomgpu_run: the last step's realvertvl + wr exchange are still to come (beside the next step's external substeps)
  int side_pending;          // work on the side stream that the main stream has not waited for yet (ev_side ends it)
  pom_blkcon con;            // host copy of blkcon (iint, iext, error_status live here)
  int lramp;
  double *rec_t[POMGPU_MAXREC + 1], *rec_s[POMGPU_MAXREC + 1];
  const double *host2, *host3;   // host bases registered by pomgpu_bind_host
  void (*exch)(void *, double *const *, const int *, int);
  void *exch_user;
  void (*order)(void *, const double *, int, const double *, int, double *, double *);   // pomgpu_order_fn
  void *order_user;
  double *ord_send[2], *ord_recv[2];   // [0] east/west: (kb+1) x jml, [1] north/south: (kb+1) x iml
  double *alt2[POMGPU_NGEN]; // second buffer set of ua, va, d, el, elb, uab, vab (fused external step)
  double *alt3[POMGPU_NGEN]; // third set: the intermediate generation on the rim and next to it when two substeps share a pass (k_ext_march2); allocated on first use
  int ext_parity;            // 1 while the current generation of those five lives in alt2
  int ext_deferred;          // pomgpu_mode_external was called for this (odd) substep and waits for its partner: the two run as one pass (k_ext_march2)
  // taurstrb / taurstrf (index 0 / 1): known to hold one value everywhere restore_interior looks, because the library wrote
  // it itself ("taurstrf = 1./trst", bounds_forcing.f:1043,1065; the shift :1054-1056 hands f's value to b).  k_ts_update then
  // forms taurstr from the two scalars instead of reading two 3-D arrays.  Any upload of those arrays ends the knowledge.
  int tau_known[2];
  double tau_val[2];
  // rho's round trip through rho - rmean (baropg, solver.f:854,937) has not been stored for levels 1..kbm1: inside
  // pomgpu_advance its only reader before dens rewrites rho is k_profq, which applies it to what it loads
  int rho_rt_pending;
  int rst_pending;           // trstr/srstr/taurstr of the last step exist only as (rst_fold, rst_fnew) weights
  double rst_fold, rst_fnew;
  double *d_vel;             // device: vamax, then (imax,jmax) as two doubles' worth of ints
  int *d_err;                // device error flag
  int *d_areas;              // device flag: art, aru, arv equal their defining formulas on dx, dy (k_check_areas); areas_checked: it is current
  int areas_checked;
  double *d_stats;           // device: the seven sums of domain_stats
  double *frc_dev[3][4][2];  // forcing records on the device: [kind][slot = n % 4][field], (im,jm) each
  int frc_n[3][4];           // which record number a slot holds (0 = empty)
  double *lat_dev[4];        // lateral_bc records: the 20 arrays of one record, concatenated (slot = n % 4)
  int lat_n[4];
  int frc_on, lat_on;        // forcing / lateral records were supplied: pomgpu_advance runs surface_forcing / lateral_bc
  // profiling
  bool prof_on;
  ProfEntry prof[96];
  int nprof;
  void *prof_state;
  pomgpu_transport tp;
  pomgpu_wide wide;
  pomgpu_ctx *parent;        // the tile's context, for the extended tile (errors and profile entries go there)
  int flags;                 // POMGPU_CTX_2D: no 3-D arrays (the extended tile of the wide-halo external mode)
  unsigned *ext_bar;         // device: arrival counter + abort word of k_ext_loop's grid barrier (k_ext.hip)
  unsigned ext_bar_base;     // the counter's value when the next k_ext_loop starts (every workgroup arrives once per barrier)
  int ext_loop_off;          // a k_ext_loop launch of this context gave up at its barrier: the substeps run as launches of their own from then on
  void *io_job;              // the output / restart file being written behind the model's back (cdf_out.hip), NULL = none
  int launch_err;            // first hipError_t a kernel launch returned (0 = none); reported by the next sync / get_con
  char err[512];
};

// launch helpers -----------------------------------------------------------------------------
int pomgpu_prof_slot(pomgpu_ctx *c, const char *name);
void pomgpu_prof_pre(pomgpu_ctx *c);
void pomgpu_prof_post(pomgpu_ctx *c, int slot);
int pomgpu_fail(pomgpu_ctx *c, int code, const char *fmt, ...);   // sets error_status, fills last_error, prints; returns code
void pomgpu_launch_check(pomgpu_ctx *c, const char *name);   // a refused launch sets error_status, fills last_error, prints

#define LAUNCH(c, kern, grid, block, ...)                                        \
  do {                                                                           \
    int _s = (c)->prof_on ? pomgpu_prof_slot((c), #kern) : -1;                   \
    if (_s >= 0) pomgpu_prof_pre(c);                                             \
    hipLaunchKernelGGL(kern, grid, block, 0, (c)->cur, __VA_ARGS__);             \
    pomgpu_launch_check((c), #kern);                                             \
    if (_s >= 0) pomgpu_prof_post((c), _s);                                      \
  } while (0)

// same with an explicit profile name (template kernels: several instantiations share one name)
#define LAUNCHN(c, name, kern, grid, block, ...)                                 \
  do {                                                                           \
    int _s = (c)->prof_on ? pomgpu_prof_slot((c), name) : -1;                    \
    if (_s >= 0) pomgpu_prof_pre(c);                                             \
    hipLaunchKernelGGL(kern, grid, block, 0, (c)->cur, __VA_ARGS__);             \
    pomgpu_launch_check((c), name);                                              \
    if (_s >= 0) pomgpu_prof_post((c), _s);                                      \
  } while (0)

static inline dim3 blk2() { return dim3(64, 4, 1); }
static inline dim3 grid2(const KP &P) { return dim3((P.iml + 63) / 64, (P.jml + 3) / 4, 1); }
static inline dim3 grid3(const KP &P, int nz) { return dim3((P.iml + 63) / 64, (P.jml + 3) / 4, nz); }
// banded, XCD-aware cell launches (decoded by MARCH3 in k_adv.hip): 1-D grid, linear id
// L = 8*m + xcd, m = ((q*kb + (k-1))*bpl + p), band = 8*q + xcd, p = block inside the band.
// KP.g_rb rows per band, KP.g_nbx blocks along i, KP.g_bpl blocks per band and level.
static inline dim3 gridm(const KP &P) {
  const long nbands = (P.jml + P.g_rb - 1) / P.g_rb;
  const long rounds = (nbands + 7) / 8;
  return dim3((unsigned)(8 * rounds * P.kb * P.g_bpl), 1, 1);
}
static inline void set_band_geometry(KP &P, const pomgpu_switches &sw) {
  long budget = 3L << 20;                       // bytes of one XCD's L2 a band may occupy per level
  const bool e = sw.on[SW_BAND_BYTES] != 0;
  if (e && sw.val[SW_BAND_BYTES] >= 1024) budget = sw.val[SW_BAND_BYTES];
  long rows = budget / ((long)P.iml * 8 * 24);  // ~24 arrays (2-D coefficients + 3-D operands) in flight
  rows = (rows / 4) * 4;
  if (rows < 4) rows = 4;
  if (rows > ((P.jml + 3) / 4) * 4) rows = ((P.jml + 3) / 4) * 4;
  P.g_rb = (int)rows;
  P.g_nbx = (P.iml + 63) / 64;
  P.g_bpl = P.g_nbx * (int)(rows / 4);
  // the row-sharing column kernels (HALO_XCD_ORDER): strips about 12 workgroups wide on wide tiles
  const int nbx = ((P.iml + 61) / 62 + COL_WX - 1) / COL_WX;
  const int nstr = (nbx + 11) / 12;
  P.g_strip = nbx >= 20 ? (nbx + nstr - 1) / nstr : 0;
  if (sw.on[SW_COL_STRIP]) P.g_strip = (int)sw.val[SW_COL_STRIP];
  P.g_lin = sw.on[SW_NO_LIN] ? 0 : 1;
  // cell launches: of the band heights the L2 budget allows, the one that leaves the fewest empty band slots (8 bands per round)
  if (!e) {
    long best = -1, bslots = 0;
    for (long r = 4; r <= rows; r += 4) {
      const long nb = (P.jml + r - 1) / r, slots = (nb + 7) / 8 * 8 * r;
      if (best < 0 || slots < bslots || (slots == bslots && r > best)) { best = r; bslots = slots; }
    }
    P.g_rb = (int)best;
    P.g_bpl = P.g_nbx * (int)(best / 4);
  }
}

// kernel launchers implemented in the k_*.hip files (one per fused phase of the step)
// k_ext.hip
void launch_advave_a(pomgpu_ctx *c);
void launch_advave_b(pomgpu_ctx *c);
void launch_advave_c(pomgpu_ctx *c);
void launch_advave_fused(pomgpu_ctx *c);
void launch_advct_col(pomgpu_ctx *c, int sum2d);
void launch_advct_edge(pomgpu_ctx *c, double *to_e, double *to_n);
void launch_advct_fix(pomgpu_ctx *c, const double *from_w, const double *from_s);
void launch_advct_fix2d(pomgpu_ctx *c, int west, int south);
void launch_advt2x2_col(pomgpu_ctx *c, const double *tb, const double *t, const double *tc, double *tf, const double *sb, const double *s_,
                        const double *sc, double *sf);
void launch_advq2_col(pomgpu_ctx *c, const double *q, const double *qb, double *qf, const double *ql, const double *qlb, double *qlf, int zero_else);
void launch_advuv_col(pomgpu_ctx *c);
int launch_profuv_reg(pomgpu_ctx *c);   // 0 when kb is outside the instantiated range
void launch_advave_m2a(pomgpu_ctx *c);
void launch_advave_m2b(pomgpu_ctx *c);
void launch_vint(pomgpu_ctx *c, int only_aam, int ghost = 0);   // ghost: +1 = every cell but the ghost lines (aam2d only), -1 = the ghost lines alone
void launch_modeint_tail(pomgpu_ctx *c);
void launch_ext_elf(pomgpu_ctx *c);
void launch_ext_uvaf(pomgpu_ctx *c, int interior);
void launch_ext_update(pomgpu_ctx *c);
void launch_ext_step(pomgpu_ctx *c, const KP &Q, int store_f, int fuse_adv);
void launch_check_areas(pomgpu_ctx *c);
int launch_ext_loop(pomgpu_ctx *c, const KP &Q, int first, int last);   // 1 = launched (all substeps first..last), 0 = not applicable
int launch_ext_pair_ok(const pomgpu_switches &sw, const KP &Q);                                             // would launch_ext_pair take this tile?
int launch_ext_pair(pomgpu_ctx *c, const KP &Q, double *const *T, int store_f2);   // 1 = substeps Q.iext, Q.iext + 1 launched, 0 = not applicable
void launch_copy2(pomgpu_ctx *c, double *dst, const double *src);
void launch_lat(pomgpu_ctx *c, int phase, const double *rec, double fold, double fnew);   // phase 0 load, 1 shift, 2 interpolate
void launch_frc_load(pomgpu_ctx *c, const double *ra, const double *rb, double *xf, double *yf);
void launch_frc_interp(pomgpu_ctx *c, double fold, double fnew, double *x, const double *xb, const double *xf, double *y, const double *yb,
                       const double *yf);
void launch_int_tail(pomgpu_ctx *c);
void launch_bcond1(pomgpu_ctx *c);
// k_adv.hip
void launch_advct_a(pomgpu_ctx *c);
void launch_advct_b(pomgpu_ctx *c);
void launch_advct_c(pomgpu_ctx *c);
void launch_aam(pomgpu_ctx *c);
void launch_roundtrip(pomgpu_ctx *c, double *a, const double *b, int fix_kb);
void launch_roundtrip_level(pomgpu_ctx *c, double *a, const double *b, int k);
void launch_advq_flux(pomgpu_ctx *c, const double *q, const double *qb, double *xf, double *yf);
void launch_advq_step(pomgpu_ctx *c, const double *q, const double *qb, double *qf, const double *xf, const double *yf, int zero_else);
void launch_q_filter(pomgpu_ctx *c, int mask);
void launch_q_filter_rim(pomgpu_ctx *c);
void launch_mask_q(pomgpu_ctx *c);
void launch_advt1(pomgpu_ctx *c, double *fb, double *f, const double *fclim, double *ff);
void launch_copy_kb(pomgpu_ctx *c, double *f, double *g = NULL);   // level kb = level kbm1, of one or two arrays
void launch_advt2_mass(pomgpu_ctx *c);
void launch_advt2_step(pomgpu_ctx *c, const double *fbmem, const double *f, const double *eta, double *ff, int itera);
void launch_mask3(pomgpu_ctx *c, double *a, const double *m2);
void launch_smol(pomgpu_ctx *c, const double *ff);
void launch_copy3(pomgpu_ctx *c, double *dst, const double *src);
void launch_advt2_diff(pomgpu_ctx *c, const double *fb, const double *fc, double *ff);
void launch_ts_update(pomgpu_ctx *c, double fold, double fnew, int rt, int store_rst);   // uses c->tau_known / tau_val
void launch_restore_fields(pomgpu_ctx *c, double fold, double fnew);
void launch_mask_ts(pomgpu_ctx *c);
void launch_mask_uv(pomgpu_ctx *c);
void launch_mask_w(pomgpu_ctx *c);
void launch_restore(pomgpu_ctx *c, double fold, double fnew);
void launch_restore_shift(pomgpu_ctx *c);
void launch_restore_load(pomgpu_ctx *c, const double *tr, const double *sr, double tau);
void launch_dens(pomgpu_ctx *c, const double *si, const double *ti, double *rhoo);
void launch_realvertvl(pomgpu_ctx *c);
// k_vert.hip
void launch_baropg(pomgpu_ctx *c, int sum2d);
void launch_baropg_mcc(pomgpu_ctx *c, int sum2d);
void launch_order_pack(pomgpu_ctx *c, double *send_e, double *send_n);
void launch_int_uvmean(pomgpu_ctx *c);
void launch_vertvl(pomgpu_ctx *c, int mask);
void launch_profq_bc(pomgpu_ctx *c);
void launch_profq_prod(pomgpu_ctx *c, int lines_only, int rho_rt = 0);
void launch_profq(pomgpu_ctx *c, int fuse_prod, int fuse_filter, int rho_rt = 0, int jfirst = 0, int jlast = -1);   // rows jfirst..jlast (default: all)
void launch_proft(pomgpu_ctx *c, double *f, const double *wfsurf, const double *fsurf, int nbc);
int launch_proft2(pomgpu_ctx *c, double *f0, const double *wfsurf0, const double *fsurf0, int nbc0, double *f1, const double *wfsurf1, const double *fsurf1, int nbc1);
void launch_advu_profu(pomgpu_ctx *c, int do_adv, int do_prof);
void launch_advv_profv(pomgpu_ctx *c, int do_adv, int do_prof);
void launch_uv_filter(pomgpu_ctx *c, int own = 0);   // own: the ghost lines are left to the exchange (k_vert.hip)
// k_tile.hip
void launch_coef_static(pomgpu_ctx *c);
void launch_coef_dt(pomgpu_ctx *c);
void launch_coef_eta(pomgpu_ctx *c);
void launch_advt2_rows(pomgpu_ctx *c, const double *fb, const double *f, const double *fc, double *ff);
void launch_advq_col(pomgpu_ctx *c, const double *q, const double *qb, double *qf, int zero_else);
// k_bc.hip
void launch_bcond4_edges(pomgpu_ctx *c);
void launch_bcond6_edges(pomgpu_ctx *c);
void launch_bcondorl3(pomgpu_ctx *c);
int launch_halo_pack(pomgpu_ctx *c, double *const *dev, const int *nz, int count, int dir, double *to_lo, double *to_hi);
int launch_halo_pack8(pomgpu_ctx *c, double *const *dev, const int *nz, int count, double *const *to);
int launch_halo_unpack8(pomgpu_ctx *c, double *const *dev, const int *nz, int count, const double *const *from);
int launch_halo_unpack(pomgpu_ctx *c, double *const *dev, const int *nz, int count, int dir, const double *from_lo, const double *from_hi);
void launch_rect_jobs(pomgpu_ctx *c, const RectJob *jobs_dev, int njobs, int max_ni, int max_nj);
// transport.hip
int pomgpu_tp_move(pomgpu_ctx *c, const size_t *scount, const size_t *rcount);   // send[d] -> neighbour d, recv[d] <- neighbour d
int pomgpu_tp_move_ptr(pomgpu_ctx *c, const double *const *send, const size_t *scount, double *const *recv, const size_t *rcount);
int pomgpu_tp_setup(pomgpu_ctx *c, const int *nbr8);
int pomgpu_tp_rccl(pomgpu_ctx *c, const void *id128, int rank, int nranks, const char *librccl_path);
void pomgpu_tp_free(pomgpu_ctx *c);
int pomgpu_tp_reserve(pomgpu_ctx *c, const size_t *need);                       // grow the staging buffers
int pomgpu_tp_reserve2(pomgpu_ctx *c, const size_t *need);                      // ... those of the side stream
int pomgpu_materialize(pomgpu_ctx *c);                                          // every lazily kept array up to date in the mirrors (pomgpu_api.hip)
int pomgpu_side_stream(pomgpu_ctx *c);                                          // create the side stream and its events (pomgpu_api.hip); 1 = there
int pomgpu_tp_side_ok(pomgpu_ctx *c);                                           // can rounds run on the side stream (second communicator / callback mover)?
int pomgpu_tp_move_side(pomgpu_ctx *c, const size_t *scount, const size_t *rcount);   // send2 / recv2, on c->side
// k_reduce.hip
void launch_check_velocity(pomgpu_ctx *c);
void launch_domain_stats(pomgpu_ctx *c, double *out_dev);
