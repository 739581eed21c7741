// ndt_device.hpp -- device-side building blocks shared by the two kernel translation units
// (ndt_kernels.hip: grid build + throughput kernels; ndt_latency.hip: the single-scan latency path).
// Wave helpers, the wave64 fold, result publication, and the per-point bodies of
// computeDerivatives / computeHessian.  Everything sits in an anonymous namespace: each unit gets
// its own copy and is free to compile it with its own flags.
#pragma once
#include "ndt_kernels.hpp"

#include <cfloat>
#include <cstdlib>
#include <cstring>

namespace ndt {

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}

// ---------------------------------------------------------------------------
// Wave64 "reduce-scatter" of up to 32 f64 accumulators per lane, VALU only.
//
// A plain butterfly costs 6 cross-lane steps per value (29 x 6 x 2 ds_bpermute for the K2
// accumulators: measured LDS-bound, the LDS pipe was busy for the whole kernel).  Instead each
// step folds PAIRS of values: the lane keeps one value of the pair (chosen by one lane-id bit),
// hands the other to its partner, and the number of live values halves:
//   32 -> 16  v_permlane32_swap   (partner l ^ 32, keep by bit 5)
//   16 ->  8  v_permlane16_swap   (partner l ^ 16, keep by bit 4)
//    8 ->  4  DPP row_ror:8       (partner l ^ 8,  keep by bit 3 = banks 2,3)
//    4 ->  2  DPP row_half_mirror (partner l ^ 7,  keep by bit 2 = banks 1,3)
//    2 ->  1  DPP quad_perm[2,3,0,1] + v_cndmask   (partner l ^ 2, keep by bit 1)
//    final    DPP quad_perm[1,0,3,2]               (partner l ^ 1)
// ~120 VALU instructions for 29 values instead of ~520 LDS-routed ones; the order of the f64
// additions is fixed by the lane ids, so the result is deterministic.
// Afterwards lane l holds the wave total of value fold_index(l).
// ---------------------------------------------------------------------------
// Publication of one packed result row into fine-grained pinned HOST memory by lanes 0..31 of one
// wave: 31 system-scope write-through stores, drained, then the sequence word.  No cache-wide
// write-back/invalidate (a __threadfence_system() here costs a buffer_wbl2 + buffer_inv, several us).
__device__ __forceinline__ void publish_row(double* __restrict__ row, double value, unsigned long long seq) {
  if (threadIdx.x < kEvalStride - 1) __hip_atomic_store(row + threadIdx.x, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == kEvalStride - 1)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(row) + (kEvalStride - 1), seq, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// Tagged publication (single-scan paths).  Every 8-byte word that crosses PCIe carries its own
// validity tag: word = (32 payload bits << 32) | (low 32 bits of the sequence number); value k of the
// row travels as words 2k (low half) and 2k + 1 (high half).  One store instruction of 64 lanes, no
// drain between data and flag (the drained form above costs a PCIe-visible round trip per
// evaluation), and no assumption about how the 512 bytes are split into bus transactions: a lane's
// aligned 8-byte store is single-copy atomic, and the host accepts the row only when all 64 words
// carry the expected tag.  vals: kEvalStride doubles in LDS.
constexpr int kPubWords = 2 * kEvalStride;
__device__ __forceinline__ unsigned long long tag_word(unsigned payload, unsigned long long seq) {
  return (static_cast<unsigned long long>(payload) << 32) | (seq & 0xffffffffull);
}
__device__ __forceinline__ void publish_row_tagged(double* __restrict__ pub, const double* vals, int tid,
                                                   unsigned long long seq) {
  if (tid < kPubWords) {
    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(vals[tid >> 1]));
    const unsigned payload = (tid & 1) ? static_cast<unsigned>(bits >> 32) : static_cast<unsigned>(bits);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(pub) + tid, tag_word(payload, seq), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Fixed-order sum over the per-block partial rows by one workgroup of PARTS * kEvalStride threads:
// thread (part, k) adds rows part, part + PARTS, ... of column k, 16 loads (sc1: the rows were
// written through by other CUs) in flight at a time; the PARTS partial sums meet in lds2.
template <int PARTS>
__device__ __forceinline__ double sum_rows_fixed(const double* __restrict__ partials, int n_blocks, int tid) {
  const int k = tid % kEvalStride, part = tid / kEvalStride;
  double v = 0.0;
  for (int base = part; base < n_blocks; base += 16 * PARTS) {
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int b = base + j * PARTS;
      a[j] = (b < n_blocks) ? __hip_atomic_load(partials + static_cast<size_t>(b) * kEvalStride + k, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT)
                            : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) v += a[j];
  }
  return v;
}

typedef unsigned fold_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mk_f64(int lo, int hi) { return __hiloint2double(hi, lo); }

__device__ __forceinline__ double fold32(double a, double b) {  // lanes 0-31: sum_a(l, l+32); lanes 32-63: sum_b
  const fold_u2 r0 = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(__double2loint(a)), static_cast<unsigned>(__double2loint(b)), false, false);
  const fold_u2 r1 = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(__double2hiint(a)), static_cast<unsigned>(__double2hiint(b)), false, false);
  return mk_f64(r0[0], r1[0]) + mk_f64(r0[1], r1[1]);
}
__device__ __forceinline__ double fold16(double a, double b) {  // even rows: sum_a(l, l+16); odd rows: sum_b
  const fold_u2 r0 = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2loint(a)), static_cast<unsigned>(__double2loint(b)), false, false);
  const fold_u2 r1 = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2hiint(a)), static_cast<unsigned>(__double2hiint(b)), false, false);
  return mk_f64(r0[0], r1[0]) + mk_f64(r0[1], r1[1]);
}
// lanes whose DPP bank is in BANK_B keep b, the others keep a; the partner (permutation CTRL,
// which must flip the selecting lane bit) supplies its copy of the kept value
template <int CTRL, int BANK_B>
__device__ __forceinline__ double fold_dpp(double a, double b) {
  const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const int own_lo = __builtin_amdgcn_update_dpp(alo, blo, 0xE4, 0xF, BANK_B, false);
  const int own_hi = __builtin_amdgcn_update_dpp(ahi, bhi, 0xE4, 0xF, BANK_B, false);
  const int oth_lo = __builtin_amdgcn_update_dpp(blo, alo, 0xE4, 0xF, BANK_B, false);
  const int oth_hi = __builtin_amdgcn_update_dpp(bhi, ahi, 0xE4, 0xF, BANK_B, false);
  const int p_lo = __builtin_amdgcn_update_dpp(0, oth_lo, CTRL, 0xF, 0xF, false);
  const int p_hi = __builtin_amdgcn_update_dpp(0, oth_hi, CTRL, 0xF, 0xF, false);
  return mk_f64(own_lo, own_hi) + mk_f64(p_lo, p_hi);
}
template <int CTRL>
__device__ __forceinline__ double fold_sel(double a, double b, bool keep_b) {
  const double own = keep_b ? b : a, oth = keep_b ? a : b;
  const int p_lo = __builtin_amdgcn_update_dpp(0, __double2loint(oth), CTRL, 0xF, 0xF, false);
  const int p_hi = __builtin_amdgcn_update_dpp(0, __double2hiint(oth), CTRL, 0xF, 0xF, false);
  return own + mk_f64(p_lo, p_hi);
}
__device__ __forceinline__ int fold_index(int lane) {
  return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) |
         (((lane >> 1) & 1) << 4);
}

template <int NV>
__device__ __forceinline__ double wave_fold(const double (&acc)[NV]) {
  static_assert(NV >= 1 && NV <= 32, "wave_fold handles up to 32 values");
  constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
  const int lane = threadIdx.x & (kWave - 1);
  double r1[16], r2[8], r3[4], r4[2];
#pragma unroll
  for (int i = 0; i < 16; i++) r1[i] = (i < N1) ? fold32(acc[(2 * i < NV) ? 2 * i : 0], (2 * i + 1 < NV) ? acc[(2 * i + 1 < NV) ? 2 * i + 1 : 0] : 0.0) : 0.0;
#pragma unroll
  for (int i = 0; i < 8; i++) r2[i] = (i < N2) ? fold16(r1[2 * i], (2 * i + 1 < N1) ? r1[2 * i + 1] : 0.0) : 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) r3[i] = (i < N3) ? fold_dpp<0x128, 0xC>(r2[2 * i], (2 * i + 1 < N2) ? r2[2 * i + 1] : 0.0) : 0.0;
#pragma unroll
  for (int i = 0; i < 2; i++) r4[i] = (i < N4) ? fold_dpp<0x141, 0xA>(r3[2 * i], (2 * i + 1 < N3) ? r3[2 * i + 1] : 0.0) : 0.0;
  const double r5 = fold_sel<0x4E>(r4[0], (N4 > 1) ? r4[1] : 0.0, (lane & 2) != 0);
  const int q_lo = __builtin_amdgcn_update_dpp(0, __double2loint(r5), 0xB1, 0xF, 0xF, false);
  const int q_hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r5), 0xB1, 0xF, 0xF, false);
  return r5 + mk_f64(q_lo, q_hi);
}

// Block-level sum of NV (<= 32) doubles per thread -> out[0..NV).  Fixed fold + fixed wave order:
// deterministic.  lds: [kBlock / kWave][32] doubles.
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NV], double* __restrict__ out, double* lds) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const double tot = wave_fold<NV>(acc);
  if ((lane & 1) == 0) lds[wave * 32 + fold_index(lane)] = tot;
  __syncthreads();
  if (threadIdx.x < NV) {
    double v = lds[threadIdx.x];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; w++) v += lds[w * 32 + threadIdx.x];
    out[threadIdx.x] = v;
  }
}

constexpr int kLutRejected = 0x40000000;  // LUT flag: voxel has a record but nr_points == -1

__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

// XCD-aware work assignment for the latency kernels.  Workgroups are dispatched round-robin over the 8
// XCDs (block b runs on XCD b % 8), each XCD has its own L2, and the source scan is in lattice-cell
// order: giving XCD x the x-th contiguous eighth of the scan keeps every L2 working on one compact
// region of the target's voxel records instead of all eight caching the whole map.  Returns the chunk
// (of `TPB` consecutive points) that block b of nb works on; a permutation of 0..nb-1 whatever the
// real block -> XCD placement is, so only speed depends on it.
__device__ __forceinline__ int xcd_chunk(int b, int nb) {
  const int x = b & 7, q = nb >> 3, r = nb & 7;
  return x * q + min(x, r) + (b >> 3);
}

// ---------------------------------------------------------------------------
// K2  derivatives
// ---------------------------------------------------------------------------
// neighbour offsets: DIRECT7 order of getNeighborhoodAtPoint7 (_impl.hpp:423-430);
// DIRECT26 = [PCL] getAllNeighborCellIndices(): 13 "half" offsets then their negatives.
__device__ __constant__ signed char kOff7[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
__device__ __constant__ signed char kOff26[26][3] = {
    {-1, -1, -1}, {-1, 0, -1}, {-1, 1, -1}, {0, -1, -1}, {0, 0, -1}, {0, 1, -1}, {1, -1, -1}, {1, 0, -1}, {1, 1, -1},
    {-1, -1, 0},  {0, -1, 0},  {1, -1, 0},  {-1, 0, 0},
    {1, 1, 1},    {1, 0, 1},   {1, -1, 1},  {0, 1, 1},   {0, 0, 1},  {0, -1, 1}, {-1, 1, 1},  {-1, 0, 1}, {-1, -1, 1},
    {1, 1, 0},    {0, 1, 0},   {-1, 1, 0},  {1, 0, 0}};

template <int NNB>
__device__ __forceinline__ void nb_offset(int k, int& dx, int& dy, int& dz) {
  if (NNB == 27) { dx = k / 9 - 1; dy = (k / 3) % 3 - 1; dz = k % 3 - 1; }  // KDTREE: the 3x3x3 block
  else if (NNB == 26) { dx = kOff26[k][0]; dy = kOff26[k][1]; dz = kOff26[k][2]; }
  else { dx = kOff7[k][0]; dy = kOff7[k][1]; dz = kOff7[k][2]; }
}

// [PCL 1.10] Transformer<float>::se3: x*c0 + (y*c1 + (z*c2 + c3)), f32, unfused.
__device__ __forceinline__ void xform_point(const float* T, float x, float y, float z, float& ox, float& oy, float& oz) {
  // Plain operators lexically inside contract(off): every product and sum is rounded to f32 on its
  // own, like the SSE code of the reference build (the __f*_rn wrappers would be inlined with the
  // translation unit's default contract(fast) and fuse).
#pragma clang fp contract(off)
  const float ax = z * T[2], ay = z * T[6], az = z * T[10];
  const float bx = ax + T[3], by = ay + T[7], bz = az + T[11];
  const float cx = y * T[1], cy = y * T[5], cz = y * T[9];
  const float dx = cx + bx, dy = cy + by, dz = cz + bz;
  const float ex = x * T[0], ey = x * T[4], ez = x * T[8];
  ox = ex + dx;
  oy = ey + dy;
  oz = ez + dz;
}

// voxel coordinate while SEARCHING: floor(x / leaf), _impl.hpp:379-381 (division, trap 2)
__device__ __forceinline__ void search_ijk(const GridGeom& g, float x, float y, float z, int& i, int& j, int& k) {
  i = static_cast<int>(floorf(__fdiv_rn(x, g.leaf[0])));
  j = static_cast<int>(floorf(__fdiv_rn(y, g.leaf[1])));
  k = static_cast<int>(floorf(__fdiv_rn(z, g.leaf[2])));
}

// record index of voxel (i+dx, j+dy, k+dz) or -1  (_impl.hpp:382-399)
__device__ __forceinline__ int probe(const GridView& gv, int i, int j, int k, int dx, int dy, int dz) {
  const int ci = i + dx, cj = j + dy, ck = k + dz;
  if (ci < gv.g.min_b[0] || ci > gv.g.max_b[0] || cj < gv.g.min_b[1] || cj > gv.g.max_b[1] || ck < gv.g.min_b[2] ||
      ck > gv.g.max_b[2])
    return -1;
  const int cell = (ci - gv.g.min_b[0]) * gv.g.mul[0] + (cj - gv.g.min_b[1]) * gv.g.mul[1] + (ck - gv.g.min_b[2]) * gv.g.mul[2];
  const int e = gv.lut[cell];
  return (e & kLutRejected) ? -1 : e;  // -1 has the bit set too
}

// KDTREE: record index of voxel (i+dx, ...) if it is in the centroid cloud (valid or rejected) and
// its f32 centroid is closer than the radius -- radiusSearch, voxel_grid_covariance_omp.h:476-505,
// [FLANN] L2_Simple accumulated in f32, RadiusResultSet keeps dist < r^2.  Else -1.
__device__ __forceinline__ int probe_kd(const GridView& gv, int i, int j, int k, int dx, int dy, int dz, float x, float y,
                                        float z, float r2) {
  const int ci = i + dx, cj = j + dy, ck = k + dz;
  if (ci < gv.g.min_b[0] || ci > gv.g.max_b[0] || cj < gv.g.min_b[1] || cj > gv.g.max_b[1] || ck < gv.g.min_b[2] ||
      ck > gv.g.max_b[2])
    return -1;
  const int cell = (ci - gv.g.min_b[0]) * gv.g.mul[0] + (cj - gv.g.min_b[1]) * gv.g.mul[1] + (ck - gv.g.min_b[2]) * gv.g.mul[2];
  const int e = gv.lut[cell];
  if (e < 0) return -1;
  const int rix = e & ~kLutRejected;
  const float4 c = reinterpret_cast<const float4*>(gv.recs + rix)[3];  // centroid x,y,z, n
  float d;
  {
#pragma clang fp contract(off)
    const float ex = x - c.x, ey = y - c.y, ez = z - c.z;
    d = ex * ex;
    d = d + ey * ey;
    d = d + ez * ez;
  }
  return (d < r2) ? rix : -1;
}

// coarse reject so that i+d cannot overflow and far-away points cost nothing
__device__ __forceinline__ bool near_grid(const GridGeom& g, int i, int j, int k) {
  return i >= g.min_b[0] - 1 && i <= g.max_b[0] + 1 && j >= g.min_b[1] - 1 && j <= g.max_b[1] + 1 && k >= g.min_b[2] - 1 &&
         k <= g.max_b[2] + 1;
}

struct RecRegs {
  double mx, my, mz;
  float c00, c01, c02, c11, c12, c22;
};
__device__ __forceinline__ RecRegs load_rec(const VoxelRec* __restrict__ recs, int r) {
  const float4* p = reinterpret_cast<const float4*>(recs + r);
  const float4 a = p[0], b = p[1], c = p[2];
  RecRegs o;
  o.mx = __hiloint2double(__float_as_int(a.y), __float_as_int(a.x));
  o.my = __hiloint2double(__float_as_int(a.w), __float_as_int(a.z));
  o.mz = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x));
  o.c00 = b.z; o.c01 = b.w; o.c02 = c.x; o.c11 = c.y; o.c12 = c.z; o.c22 = c.w;
  return o;
}

// per-point pieces of computePointDerivatives (f32, ndt_omp_impl.hpp:398-440):
// xj = j_ang * x (8), xh = h_ang * x (15)
struct PointDeriv {
  float j[8];
  float h[15];
};
template <class P>
__device__ __forceinline__ void point_derivatives(const P& prm, float x, float y, float z, PointDeriv& d, bool want_h) {
#pragma unroll
  for (int r = 0; r < 8; r++) d.j[r] = (prm.j[r][0] * x + prm.j[r][1] * y) + prm.j[r][2] * z;
  if (want_h) {
#pragma unroll
    for (int r = 0; r < 15; r++) d.h[r] = (prm.h[r][0] * x + prm.h[r][1] * y) + prm.h[r][2] * z;
  }
}

// updateDerivatives (ndt_omp_impl.hpp:484-537) for one (point, voxel) pair.
// f32 arithmetic in the reference's operation order with the structural zeros
// of J_E / H_E skipped (those products are exact zeros there); f64 accumulation.
// acc: [0]=score [1..6]=gradient [7..27]=Hessian upper triangle [28]=neighbour count
template <bool WANT_H>
__device__ __forceinline__ void accumulate_neighbor(double (&acc)[kNumAcc], const PointDeriv& d, float x0, float x1,
                                                    float x2, const RecRegs& r, double d1, float d2) {
  // xc = x'^T C   (x_trans4 * c_inv4)
  const float xc0 = (x0 * r.c00 + x1 * r.c01) + x2 * r.c02;
  const float xc1 = (x0 * r.c01 + x1 * r.c11) + x2 * r.c12;
  const float xc2 = (x0 * r.c02 + x1 * r.c12) + x2 * r.c22;
  const float q = (x0 * xc0 + x1 * xc1) + x2 * xc2;
  float e = expf(-d2 * q * 0.5f);                              // :499
  const float score_inc = static_cast<float>(-d1 * static_cast<double>(e));  // :501
  e = d2 * e;                                                  // :503
  if (e > 1.0f || e < 0.0f || e != e) return;                  // :506-507 (adds nothing, not even the score)
  e = static_cast<float>(static_cast<double>(e) * d1);         // :510
  acc[0] += static_cast<double>(score_inc);
  acc[28] += 1.0;

  // CJ = C * J_E columns 3..5 (columns 0..2 are the columns of C)
  const float* j = d.j;
  const float cj03 = r.c01 * j[0] + r.c02 * j[1], cj13 = r.c11 * j[0] + r.c12 * j[1], cj23 = r.c12 * j[0] + r.c22 * j[1];
  const float cj04 = (r.c00 * j[2] + r.c01 * j[3]) + r.c02 * j[4];
  const float cj14 = (r.c01 * j[2] + r.c11 * j[3]) + r.c12 * j[4];
  const float cj24 = (r.c02 * j[2] + r.c12 * j[3]) + r.c22 * j[4];
  const float cj05 = (r.c00 * j[5] + r.c01 * j[6]) + r.c02 * j[7];
  const float cj15 = (r.c01 * j[5] + r.c11 * j[6]) + r.c12 * j[7];
  const float cj25 = (r.c02 * j[5] + r.c12 * j[6]) + r.c22 * j[7];
  // g = x'^T CJ
  float g[6];
  g[0] = xc0; g[1] = xc1; g[2] = xc2;
  g[3] = (x0 * cj03 + x1 * cj13) + x2 * cj23;
  g[4] = (x0 * cj04 + x1 * cj14) + x2 * cj24;
  g[5] = (x0 * cj05 + x1 * cj15) + x2 * cj25;
#pragma unroll
  for (int k = 0; k < 6; k++) acc[1 + k] += static_cast<double>(e * g[k]);  // :515

  if (WANT_H) {
    // JCJ(b,a) = J_E[:,b] . CJ[:,a]   needed for a <= b
    const float CJ[3][6] = {{r.c00, r.c01, r.c02, cj03, cj04, cj05},
                            {r.c01, r.c11, r.c12, cj13, cj14, cj15},
                            {r.c02, r.c12, r.c22, cj23, cj24, cj25}};
    // x'^T C H_E blocks (symmetric 3x3 in the angle indices): a b c / b d e / c e f
    const float* h = d.h;
    const float xa = xc1 * h[0] + xc2 * h[1];
    const float xb = xc1 * h[2] + xc2 * h[3];
    const float xcc = xc1 * h[4] + xc2 * h[5];
    const float xd = (xc0 * h[6] + xc1 * h[7]) + xc2 * h[8];
    const float xe = (xc0 * h[9] + xc1 * h[10]) + xc2 * h[11];
    const float xf = (xc0 * h[12] + xc1 * h[13]) + xc2 * h[14];
    const float xH[3][3] = {{xa, xb, xcc}, {xb, xd, xe}, {xcc, xe, xf}};
    int idx = 7;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
      for (int jj = i; jj < 6; jj++) {
        // JCJ(jj, i)
        float jcj;
        if (jj < 3) jcj = CJ[jj][i];
        else if (jj == 3) jcj = j[0] * CJ[1][i] + j[1] * CJ[2][i];
        else if (jj == 4) jcj = (j[2] * CJ[0][i] + j[3] * CJ[1][i]) + j[4] * CJ[2][i];
        else jcj = (j[5] * CJ[0][i] + j[6] * CJ[1][i]) + j[7] * CJ[2][i];
        const float xh = (i >= 3) ? xH[i - 3][jj - 3] : 0.0f;
        const float term = e * (((-d2 * g[i]) * g[jj] + xh) + jcj);  // :529-531
        acc[idx++] += static_cast<double>(term);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Factored form of updateDerivatives.  With xc = C x' (C symmetric) the reference's per-neighbour
// quantities are
//   gradient_k     += e * (xc . J_k)
//   hessian(i,j)   += e * ( -d2 (xc . J_i)(xc . J_j) + xc . HE_ij + J_j^T C J_i )
// and J_k, HE_ij depend on the POINT only.  Everything is therefore linear in two small
// per-neighbour objects,
//   xe = sum_n e_n xc_n                      (3)
//   A  = sum_n e_n (C_n - d2 xc_n xc_n^T)    (3x3 symmetric, 6)
// which are accumulated over the <= 7 neighbours of a point (f32), after which
//   gradient = J^T xe ,  hessian = J^T A J + [xe . HE_ij]   are formed ONCE per point and added to
// the f64 accumulators.  ~60 instead of ~350 VALU instructions per neighbour; the f32 rounding of
// the per-point sums differs from the reference's per-neighbour rounding by O(1e-7) relative per
// point (same order as its own f32 noise), far inside the parity tolerance.
// ---------------------------------------------------------------------------
struct PointAcc {
  float xe0, xe1, xe2;
  float a00, a01, a02, a11, a12, a22;
};

template <bool WANT_H>
__device__ __forceinline__ void accumulate_neighbor_factored(double& score, double& nn, PointAcc& pa, float x0, float x1,
                                                             float x2, const RecRegs& r, double d1, float d2) {
  const float xc0 = (x0 * r.c00 + x1 * r.c01) + x2 * r.c02;
  const float xc1 = (x0 * r.c01 + x1 * r.c11) + x2 * r.c12;
  const float xc2 = (x0 * r.c02 + x1 * r.c12) + x2 * r.c22;
  const float q = (x0 * xc0 + x1 * xc1) + x2 * xc2;
  float e = expf(-d2 * q * 0.5f);                                            // :499
  const float score_inc = static_cast<float>(-d1 * static_cast<double>(e));  // :501
  e = d2 * e;                                                                // :503
  if (e > 1.0f || e < 0.0f || e != e) return;                                // :506-507
  e = static_cast<float>(static_cast<double>(e) * d1);                       // :510
  score += static_cast<double>(score_inc);
  nn += 1.0;
  pa.xe0 += e * xc0;
  pa.xe1 += e * xc1;
  pa.xe2 += e * xc2;
  if (WANT_H) {
    const float t0 = (-d2 * e) * xc0, t1 = (-d2 * e) * xc1, t2 = (-d2 * e) * xc2;
    pa.a00 += e * r.c00 + t0 * xc0;
    pa.a01 += e * r.c01 + t0 * xc1;
    pa.a02 += e * r.c02 + t0 * xc2;
    pa.a11 += e * r.c11 + t1 * xc1;
    pa.a12 += e * r.c12 + t1 * xc2;
    pa.a22 += e * r.c22 + t2 * xc2;
  }
}

// J_E = [ I3 | B ],  B columns: (0, j0, j1), (j2, j3, j4), (j5, j6, j7)   (ndt_omp_impl.hpp:407-414)
template <bool WANT_H>
__device__ __forceinline__ void finish_point(double (&acc)[kNumAcc], const PointAcc& pa, const PointDeriv& d) {
  const float* j = d.j;
  const float B[3][3] = {{0.0f, j[2], j[5]}, {j[0], j[3], j[6]}, {j[1], j[4], j[7]}};
  acc[1] += static_cast<double>(pa.xe0);
  acc[2] += static_cast<double>(pa.xe1);
  acc[3] += static_cast<double>(pa.xe2);
  acc[4] += static_cast<double>(pa.xe1 * B[1][0] + pa.xe2 * B[2][0]);
  acc[5] += static_cast<double>((pa.xe0 * B[0][1] + pa.xe1 * B[1][1]) + pa.xe2 * B[2][1]);
  acc[6] += static_cast<double>((pa.xe0 * B[0][2] + pa.xe1 * B[1][2]) + pa.xe2 * B[2][2]);
  if (WANT_H) {
    const float A[3][3] = {{pa.a00, pa.a01, pa.a02}, {pa.a01, pa.a11, pa.a12}, {pa.a02, pa.a12, pa.a22}};
    float AB[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      AB[r][0] = A[r][1] * B[1][0] + A[r][2] * B[2][0];
#pragma unroll
      for (int c = 1; c < 3; c++) AB[r][c] = (A[r][0] * B[0][c] + A[r][1] * B[1][c]) + A[r][2] * B[2][c];
    }
    // x-block of H_E: a b c / b d e / c e f  with a=(0,h0,h1) b=(0,h2,h3) c=(0,h4,h5) d=(h6,h7,h8) ...
    const float* h = d.h;
    const float xa = pa.xe1 * h[0] + pa.xe2 * h[1];
    const float xb = pa.xe1 * h[2] + pa.xe2 * h[3];
    const float xcc = pa.xe1 * h[4] + pa.xe2 * h[5];
    const float xd = (pa.xe0 * h[6] + pa.xe1 * h[7]) + pa.xe2 * h[8];
    const float xe = (pa.xe0 * h[9] + pa.xe1 * h[10]) + pa.xe2 * h[11];
    const float xf = (pa.xe0 * h[12] + pa.xe1 * h[13]) + pa.xe2 * h[14];
    const float X[3][3] = {{xa, xb, xcc}, {xb, xd, xe}, {xcc, xe, xf}};
    // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
    int idx = 7;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int c = i; c < 3; c++) acc[idx++] += static_cast<double>(A[i][c]);
#pragma unroll
      for (int c = 0; c < 3; c++) acc[idx++] += static_cast<double>(AB[i][c]);
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = a; b < 3; b++) {
        float v = (a == 0) ? (B[1][0] * AB[1][b] + B[2][0] * AB[2][b])
                           : ((B[0][a] * AB[0][b] + B[1][a] * AB[1][b]) + B[2][a] * AB[2][b]);
        acc[idx++] += static_cast<double>(v + X[a][b]);
      }
  }
}

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// PRELOADED: the caller already holds src[first] (the evaluation server keeps each lane's first point in registers
// across rounds: the scan does not change between the evaluations of a registration, and the load would otherwise
// head every round's dependent chain)
template <int NNB, bool WANT_H, class P, bool STAMP = false, bool FACTORED = true, bool PRELOADED = false>
__device__ __forceinline__ void derivatives_body(const float4* __restrict__ src, int n, const GridView& gv, const P& prm,
                                                 int first, int stride, double (&acc)[kNumAcc],
                                                 unsigned long long* st = nullptr, float4 first_pt = float4{0.f, 0.f, 0.f, 0.f}) {
  for (int i = first; i < n; i += stride) {
    const float4 pt = (PRELOADED && i == first) ? first_pt : src[i];
    if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[1] = stamp(); }
    float tx, ty, tz;
    xform_point(prm.T, pt.x, pt.y, pt.z, tx, ty, tz);
    // A non-finite point has no neighbourhood in the reference (its voxel index is garbage and fails
    // the bounding-box test): it contributes nothing.  Without this, 0 x NaN of its point derivatives
    // would poison the per-point finish even though every neighbour term is rejected.
    if (!finite3(tx, ty, tz)) continue;
    int vi, vj, vk;
    search_ijk(gv.g, tx, ty, tz, vi, vj, vk);
    if (!near_grid(gv.g, vi, vj, vk)) continue;
    int rec[NNB];
    bool any = false;
#pragma unroll
    for (int k = 0; k < NNB; k++) {
      int dx, dy, dz;
      nb_offset<NNB>(k, dx, dy, dz);
      rec[k] = probe(gv, vi, vj, vk, dx, dy, dz);
      any |= (rec[k] >= 0);
    }
    if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[2] = stamp(); }
    if (!any) continue;
    PointDeriv d;
    point_derivatives(prm, pt.x, pt.y, pt.z, d, WANT_H);
    // Software pipeline over the neighbours: the record of neighbour k+1 is requested (index
    // clamped, so the load is unconditional and hoistable) before neighbour k's math runs; one
    // record gather latency is exposed per point instead of one per neighbour.
    RecRegs cur = load_rec(gv.recs, rec[0] < 0 ? 0 : rec[0]);
    if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[3] = stamp(); }
    PointAcc pa = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NNB; k++) {
      RecRegs nxt = cur;
      if (k + 1 < NNB) nxt = load_rec(gv.recs, rec[k + 1] < 0 ? 0 : rec[k + 1]);
      if (rec[k] >= 0) {
        // x_trans (f32 -> f64) - mean (f64), rounded to f32  (:259-262, :492)
        const float x0 = static_cast<float>(static_cast<double>(tx) - cur.mx);
        const float x1 = static_cast<float>(static_cast<double>(ty) - cur.my);
        const float x2 = static_cast<float>(static_cast<double>(tz) - cur.mz);
        if (FACTORED) accumulate_neighbor_factored<WANT_H>(acc[0], acc[28], pa, x0, x1, x2, cur, prm.d1, prm.d2);
        else accumulate_neighbor<WANT_H>(acc, d, x0, x1, x2, cur, prm.d1, prm.d2);
      }
      cur = nxt;
    }
    if (FACTORED) finish_point<WANT_H>(acc, pa, d);
    if (STAMP) st[4] = stamp();
  }
}

// DIRECT7, latency-oriented decomposition: one (point, neighbour) task per lane, 8 consecutive
// lanes share a point (slot 7 idles).  Seven dependent gathers per point become seven parallel
// lanes, so a 100k-point scan exposes 700k independent tasks instead of 100k serial chains.
template <bool WANT_H, class P>
__device__ __forceinline__ void derivatives_body_split7(const float4* __restrict__ src, int n, const GridView& gv,
                                                        const P& prm, int first, int stride, double (&acc)[kNumAcc]) {
  const int slot = threadIdx.x & 7;
  if (slot == 7) return;
  // order of getNeighborhoodAtPoint7 (_impl.hpp:423-430): centre, +x, -x, +y, -y, +z, -z
  const int dx = (slot == 1) - (slot == 2), dy = (slot == 3) - (slot == 4), dz = (slot == 5) - (slot == 6);
  const long long total = static_cast<long long>(n) * 8;
  for (long long t = first; t < total; t += stride) {
    const float4 pt = src[t >> 3];
    float tx, ty, tz;
    xform_point(prm.T, pt.x, pt.y, pt.z, tx, ty, tz);
    // A non-finite point has no neighbourhood in the reference (its voxel index is garbage and fails
    // the bounding-box test): it contributes nothing.  Without this, 0 x NaN of its point derivatives
    // would poison the per-point finish even though every neighbour term is rejected.
    if (!finite3(tx, ty, tz)) continue;
    int vi, vj, vk;
    search_ijk(gv.g, tx, ty, tz, vi, vj, vk);
    if (!near_grid(gv.g, vi, vj, vk)) continue;
    const int rix = probe(gv, vi, vj, vk, dx, dy, dz);
    if (rix < 0) continue;
    const RecRegs r = load_rec(gv.recs, rix);
    PointDeriv d;
    point_derivatives(prm, pt.x, pt.y, pt.z, d, WANT_H);
    const float x0 = static_cast<float>(static_cast<double>(tx) - r.mx);
    const float x1 = static_cast<float>(static_cast<double>(ty) - r.my);
    const float x2 = static_cast<float>(static_cast<double>(tz) - r.mz);
    accumulate_neighbor<WANT_H>(acc, d, x0, x1, x2, r, prm.d1, prm.d2);
  }
}

// KDTREE search: 3x3x3 cells around the point, centroid-distance filter, same factored math.
// (The reference visits the hits sorted by distance; only the f64 summation order differs.)
template <bool WANT_H, class P>
__device__ __forceinline__ void derivatives_body_kd(const float4* __restrict__ src, int n, const GridView& gv, const P& prm,
                                                    int first, int stride, double (&acc)[kNumAcc]) {
  const float r2 = __int_as_float(prm.pad);
  for (int i = first; i < n; i += stride) {
    const float4 pt = src[i];
    float tx, ty, tz;
    xform_point(prm.T, pt.x, pt.y, pt.z, tx, ty, tz);
    // A non-finite point has no neighbourhood in the reference (its voxel index is garbage and fails
    // the bounding-box test): it contributes nothing.  Without this, 0 x NaN of its point derivatives
    // would poison the per-point finish even though every neighbour term is rejected.
    if (!finite3(tx, ty, tz)) continue;
    int vi, vj, vk;
    search_ijk(gv.g, tx, ty, tz, vi, vj, vk);
    if (!near_grid(gv.g, vi, vj, vk)) continue;
    PointAcc pa = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const double nn0 = acc[28];
    for (int a = -1; a <= 1; a++)
      for (int b = -1; b <= 1; b++)
        for (int c = -1; c <= 1; c++) {
          const int rix = probe_kd(gv, vi, vj, vk, a, b, c, tx, ty, tz, r2);
          if (rix < 0) continue;
          const RecRegs r = load_rec(gv.recs, rix);
          const float x0 = static_cast<float>(static_cast<double>(tx) - r.mx);
          const float x1 = static_cast<float>(static_cast<double>(ty) - r.my);
          const float x2 = static_cast<float>(static_cast<double>(tz) - r.mz);
          accumulate_neighbor_factored<WANT_H>(acc[0], acc[28], pa, x0, x1, x2, r, prm.d1, prm.d2);
        }
    if (acc[28] != nn0) {
      PointDeriv d;
      point_derivatives(prm, pt.x, pt.y, pt.z, d, WANT_H);
      finish_point<WANT_H>(acc, pa, d);
    }
  }
}

// ---------------------------------------------------------------------------
// computeHessian / updateHessian, all f64 (ndt_omp_impl.hpp:540-645, 443-481)
// acc layout identical to k_derivatives (only [7..27] are written).
// ---------------------------------------------------------------------------
// Factored like the f32 path (see accumulate_neighbor_factored), everything in f64:
//   per neighbour  xe += e xc ,  A += e (C - d2 xc xc^T)      with xc = C x'
//   per point      H  = J^T A J + [xe . HE_ij]                 (f64 angle vectors, -sy in d1)
struct PointAcc64 {
  double xe0, xe1, xe2;
  double a00, a01, a02, a11, a12, a22;
};

template <class P>
__device__ __forceinline__ void finish_point64(double (&acc)[kNumAcc], const PointAcc64& pa, const P& prm, double px,
                                               double py, double pz) {
  auto dot = [](const double a[3], double b0, double b1, double b2) { return (a[0] * b0 + a[1] * b1) + a[2] * b2; };
  double j[8], h[15];
#pragma unroll
  for (int r = 0; r < 8; r++) j[r] = dot(prm.jd[r], px, py, pz);
#pragma unroll
  for (int r = 0; r < 15; r++) h[r] = dot(prm.hd[r], px, py, pz);
  const double B[3][3] = {{0.0, j[2], j[5]}, {j[0], j[3], j[6]}, {j[1], j[4], j[7]}};
  const double A[3][3] = {{pa.a00, pa.a01, pa.a02}, {pa.a01, pa.a11, pa.a12}, {pa.a02, pa.a12, pa.a22}};
  double AB[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) AB[r][c] = (A[r][0] * B[0][c] + A[r][1] * B[1][c]) + A[r][2] * B[2][c];
  const double xa = pa.xe1 * h[0] + pa.xe2 * h[1];
  const double xb = pa.xe1 * h[2] + pa.xe2 * h[3];
  const double xcc = pa.xe1 * h[4] + pa.xe2 * h[5];
  const double xd = (pa.xe0 * h[6] + pa.xe1 * h[7]) + pa.xe2 * h[8];
  const double xe = (pa.xe0 * h[9] + pa.xe1 * h[10]) + pa.xe2 * h[11];
  const double xf = (pa.xe0 * h[12] + pa.xe1 * h[13]) + pa.xe2 * h[14];
  const double X[3][3] = {{xa, xb, xcc}, {xb, xd, xe}, {xcc, xe, xf}};
  int idx = 7;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int c = i; c < 3; c++) acc[idx++] += A[i][c];
#pragma unroll
    for (int c = 0; c < 3; c++) acc[idx++] += AB[i][c];
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = a; b < 3; b++) acc[idx++] += ((B[0][a] * AB[0][b] + B[1][a] * AB[1][b]) + B[2][a] * AB[2][b]) + X[a][b];
}

// all-f64 Hessian contributions (computeHessian / updateHessian, ndt_omp_impl.hpp:584-645) of the
// points first, first + stride, ... into acc
// LATE_TABLES: a compiler-level memory fence in front of the per-point finish keeps the 69 f64 table
// entries from being loaded (and, in the evaluation server, spilled) ahead of the neighbour loop.
template <int NNB, bool LATE_TABLES = false>
__device__ __forceinline__ void hessian64_body(const float4* __restrict__ src, int n, const GridView& gv,
                                               const Hess64Params& prm, int first, int stride, double (&acc)[kNumAcc]) {
  for (int i = first; i < n; i += stride) {
    const float4 pt = src[i];
    float tx, ty, tz;
    xform_point(prm.T, pt.x, pt.y, pt.z, tx, ty, tz);
    // A non-finite point has no neighbourhood in the reference (its voxel index is garbage and fails
    // the bounding-box test): it contributes nothing.  Without this, 0 x NaN of its point derivatives
    // would poison the per-point finish even though every neighbour term is rejected.
    if (!finite3(tx, ty, tz)) continue;
    int vi, vj, vk;
    search_ijk(gv.g, tx, ty, tz, vi, vj, vk);
    if (!near_grid(gv.g, vi, vj, vk)) continue;
    PointAcc64 pa = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool any = false;
    for (int k = 0; k < NNB; k++) {
      int dx, dy, dz;
      nb_offset<NNB>(k, dx, dy, dz);
      const int rix = (NNB == 27) ? probe_kd(gv, vi, vj, vk, dx, dy, dz, tx, ty, tz, static_cast<float>(prm.r2))
                                  : probe(gv, vi, vj, vk, dx, dy, dz);
      if (rix < 0) continue;
      const RecRegs r = load_rec(gv.recs, rix);
      // the record keeps icov in its f32 rounding
      const double c00 = r.c00, c01 = r.c01, c02 = r.c02, c11 = r.c11, c12 = r.c12, c22 = r.c22;
      const double x0 = static_cast<double>(tx) - r.mx, x1 = static_cast<double>(ty) - r.my, x2 = static_cast<double>(tz) - r.mz;
      const double xc0 = (c00 * x0 + c01 * x1) + c02 * x2;
      const double xc1 = (c01 * x0 + c11 * x1) + c12 * x2;
      const double xc2 = (c02 * x0 + c12 * x1) + c22 * x2;
      double e = prm.d2 * exp(-prm.d2 * ((x0 * xc0 + x1 * xc1) + x2 * xc2) / 2);  // :622
      if (e > 1 || e < 0 || e != e) continue;                                      // :625-626
      e *= prm.d1;
      any = true;
      pa.xe0 += e * xc0; pa.xe1 += e * xc1; pa.xe2 += e * xc2;
      const double t0 = (-prm.d2 * e) * xc0, t1 = (-prm.d2 * e) * xc1, t2 = (-prm.d2 * e) * xc2;
      pa.a00 += e * c00 + t0 * xc0; pa.a01 += e * c01 + t0 * xc1; pa.a02 += e * c02 + t0 * xc2;
      pa.a11 += e * c11 + t1 * xc1; pa.a12 += e * c12 + t1 * xc2; pa.a22 += e * c22 + t2 * xc2;
    }
    if (LATE_TABLES) asm volatile("" ::: "memory");
    if (any) finish_point64(acc, pa, prm, pt.x, pt.y, pt.z);
  }
}

// ---------------------------------------------------------------------------
// small dense f64 algebra (grid finalize, GICP covariances / Mahalanobis matrices)
// ---------------------------------------------------------------------------
// 3x3 symmetric eigen-decomposition (cyclic Jacobi, f64); eigenvalues ascending
// in w[], eigenvectors in the columns of V.  Stands in for
// Eigen::SelfAdjointEigenSolver<Matrix3d> (_impl.hpp:275,333-335): only the
// eigenvalues and V*diag*V^-1 are consumed, both solver-independent to O(eps).
__device__ inline void eig3_jacobi(const double A_in[3][3], double w[3], double V[3][3]) {
#pragma clang fp contract(off)
  double A[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      A[i][j] = (i >= j) ? A_in[i][j] : A_in[j][i];  // lower triangle, like Eigen
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 50; sweep++) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    const double dia = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    // Jacobi converges quadratically and the eigenvalue error is O(off^2 / gap): at off <= eps/2 * dia
    // another sweep cannot change a bit of the result that is consumed
    if (off <= 1e-300 || off <= dia * 1e-16) break;
#pragma unroll
    for (int pq = 0; pq < 3; pq++) {
      const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
      const double apq = A[p][q];
      if (apq == 0.0) continue;
      // tan of the rotation angle, the root of t^2 + 2 theta t - 1 = 0 that is smaller in magnitude,
      // written without forming theta = d / (2 apq):  t = 2 apq / (d + sign(d) sqrt(d^2 + 4 apq^2))
      // (one division and two square roots per rotation instead of three and two: the f64 division
      // chains are what this one-thread-per-voxel kernel waits for)
      const double d = A[q][q] - A[p][p], b2 = 2.0 * apq;
      const double t = b2 / (d + copysign(sqrt(d * d + b2 * b2), d));
      const double c = rsqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; k++) {
        const double akp = A[k][p], akq = A[k][q];
        A[k][p] = c * akp - s * akq;
        A[k][q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; k++) {
        const double apk = A[p][k], aqk = A[q][k];
        A[p][k] = c * apk - s * aqk;
        A[q][k] = s * apk + c * aqk;
      }
      for (int k = 0; k < 3; k++) {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
      }
    }
  }
  // sort ascending (3 elements)
  double d[3] = {A[0][0], A[1][1], A[2][2]};
  int o[3] = {0, 1, 2};
  if (d[o[0]] > d[o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  if (d[o[1]] > d[o[2]]) { int t = o[1]; o[1] = o[2]; o[2] = t; }
  if (d[o[0]] > d[o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  double Vs[3][3];
  for (int j = 0; j < 3; j++) {
    w[j] = d[o[j]];
    for (int i = 0; i < 3; i++) Vs[i][j] = V[i][o[j]];
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) V[i][j] = Vs[i][j];
}

// Matrix3d::inverse() as Eigen evaluates it (cofactors, multiply by 1/det)
__device__ inline void inv3_cofactor(const double a[3][3], double r[3][3]) {
#pragma clang fp contract(off)
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a[i1][j1] * a[i2][j2] - a[i1][j2] * a[i2][j1];
  };
  const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  const double det = (c00 * a[0][0] + c10 * a[1][0]) + c20 * a[2][0];
  const double invdet = 1.0 / det;
  r[0][0] = c00 * invdet; r[0][1] = c10 * invdet; r[0][2] = c20 * invdet;
  r[1][0] = cof(0, 1) * invdet; r[1][1] = cof(1, 1) * invdet; r[1][2] = cof(2, 1) * invdet;
  r[2][0] = cof(0, 2) * invdet; r[2][1] = cof(1, 2) * invdet; r[2][2] = cof(2, 2) * invdet;
}

// [FLANN] L2_Simple<float>: f32, (dx*dx + dy*dy) + dz*dz, no contraction
__device__ __forceinline__ float dist2_f32(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}


}  // namespace
}  // namespace ndt
