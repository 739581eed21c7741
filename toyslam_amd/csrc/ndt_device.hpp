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
constexpr int kBatchPointsPerBlock = 2 * kBlock;  // lock-step batches: a block's contiguous run of a scan's points (measured 2 / 4 / 8 per thread: 6.54k / 6.40k / 6.36k reg/s on the 512-scan build)

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

// one wave publishes a row to the host without a trip through LDS: lane l < kEvalStride holds value l; word w of the row is
// half (w & 1) of value w >> 1
__device__ __forceinline__ void publish_lanes_tagged(double* __restrict__ pub, double value, int lane, unsigned long long seq) {
  const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(value));
  const unsigned lo = __shfl(static_cast<unsigned>(bits), lane >> 1, kWave), hi = __shfl(static_cast<unsigned>(bits >> 32), lane >> 1, kWave);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(pub) + lane, tag_word((lane & 1) ? hi : lo, seq), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
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
// Packed f32 arithmetic, written out by hand (VOP3P, two f32 lanes per instruction).
//
// The per-point math below is the same in both kernel translation units and in every kernel that
// includes this header: every product, sum and fused multiply-add is spelled out (no contraction left
// to the compiler, `#pragma clang fp contract(off)` wherever plain operators are used), so the f32
// results do not depend on the unit's flags or on how a kernel is scheduled -- the batch kernels and
// the single-scan kernels return the same per-point values bit for bit and differ only in the order
// of their f64 sums.  op_sel / op_sel_hi pick, per source, which half of its register pair feeds the
// low and the high lane: a value is broadcast from whichever half of a pair it lives in, for free.
//   b0lo / b0hi: source 0 is broadcast from its low / high half; sources 1, 2 are taken lane-wise.
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

// (the builtins below compile to exactly one v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 each, the broadcasts to
// op_sel bits of that instruction -- checked in the ISA; no contraction can happen across them)
__device__ __forceinline__ f2 bc_lo(f2 a) { return __builtin_shufflevector(a, a, 0, 0); }
__device__ __forceinline__ f2 bc_hi(f2 a) { return __builtin_shufflevector(a, a, 1, 1); }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_fma_b0lo(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(bc_lo(a), b, c); }
__device__ __forceinline__ f2 pk_fma_b0hi(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(bc_hi(a), b, c); }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ f2 pk_mul_b0lo(f2 a, f2 b) {
#pragma clang fp contract(off)
  return bc_lo(a) * b;
}
__device__ __forceinline__ f2 pk_mul_b0hi(f2 a, f2 b) {
#pragma clang fp contract(off)
  return bc_hi(a) * b;
}
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float fma1(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float mul1(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add1(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
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

// voxel coordinate while SEARCHING: floor(x / leaf), _impl.hpp:379-381 (division, trap 2).  For power-of-two
// leaves the product with the (then exact) reciprocal is the same f32 number as the quotient.
__device__ __forceinline__ void search_ijk(const GridGeom& g, float x, float y, float z, int& i, int& j, int& k) {
  if (g.pow2) {  // uniform
#pragma clang fp contract(off)
    i = static_cast<int>(floorf(x * g.inv_leaf[0]));
    j = static_cast<int>(floorf(y * g.inv_leaf[1]));
    k = static_cast<int>(floorf(z * g.inv_leaf[2]));
  } else {
    i = static_cast<int>(floorf(__fdiv_rn(x, g.leaf[0])));
    j = static_cast<int>(floorf(__fdiv_rn(y, g.leaf[1])));
    k = static_cast<int>(floorf(__fdiv_rn(z, g.leaf[2])));
  }
}

// within one cell of the bounding box: every neighbour cell of the 3x3x3 block then lies inside the padded
// look-up table (border kLutBorder = 2), and far-away points cost nothing
__device__ __forceinline__ bool near_grid(const GridGeom& g, int i, int j, int k) {
  return static_cast<unsigned>(i - g.min_b[0] + 1) <= static_cast<unsigned>(g.div_b[0] + 1) &&
         static_cast<unsigned>(j - g.min_b[1] + 1) <= static_cast<unsigned>(g.div_b[1] + 1) &&
         static_cast<unsigned>(k - g.min_b[2] + 1) <= static_cast<unsigned>(g.div_b[2] + 1);
}
// index of voxel (i, j, k) in the padded look-up table (valid for near_grid points and their neighbour cells)
__device__ __forceinline__ unsigned lut_index(const GridGeom& g, int i, int j, int k) {
  return static_cast<unsigned>(i - g.min_b[0] + kLutBorder) + static_cast<unsigned>(j - g.min_b[1] + kLutBorder) * static_cast<unsigned>(g.pmul[1]) +
         static_cast<unsigned>(k - g.min_b[2] + kLutBorder) * static_cast<unsigned>(g.pmul[2]);
}
__device__ __forceinline__ int lut_offset(const GridGeom& g, int dx, int dy, int dz) { return dx + dy * g.pmul[1] + dz * g.pmul[2]; }

// look-up table entry of voxel (ci, cj, ck) of a SPARSE grid: bounds test as _impl.hpp:382-392, then the hash walk
__device__ __forceinline__ int lut_entry_hash(const GridView& gv, int ci, int cj, int ck) {
  if (ci < gv.g.min_b[0] || ci > gv.g.max_b[0] || cj < gv.g.min_b[1] || cj > gv.g.max_b[1] || ck < gv.g.min_b[2] || ck > gv.g.max_b[2])
    return kLutEmpty;
  const int key = (ci - gv.g.min_b[0]) * gv.g.mul[0] + (cj - gv.g.min_b[1]) * gv.g.mul[1] + (ck - gv.g.min_b[2]) * gv.g.mul[2];
  const int2* tab = reinterpret_cast<const int2*>(gv.lut);
  const unsigned mask = (1u << gv.g.hash_bits) - 1u;
  for (unsigned h = hash_slot(key, gv.g.hash_bits);; h = (h + 1u) & mask) {
    const int2 e = tab[h];
    if (e.x == key) return e.y;
    if (e.x == -1) return kLutEmpty;
  }
}
// look-up table entry of the voxel at offset (dx, dy, dz) from voxel (vi, vj, vk) (centre = its padded dense index)
__device__ __forceinline__ int lut_entry(const GridView& gv, int vi, int vj, int vk, unsigned centre, int dx, int dy, int dz) {
  if (gv.g.hash_bits) return lut_entry_hash(gv, vi + dx, vj + dy, vk + dz);  // uniform
  return gv.lut[centre + static_cast<unsigned>(lut_offset(gv.g, dx, dy, dz))];
}

// record index (>= 0) of voxel (i+dx, j+dy, k+dz) if the DIRECT searches may use it, else negative
// (_impl.hpp:382-399: outside the box, empty, fewer than min_points_per_voxel points, or rejected)
__device__ __forceinline__ int probe(const GridView& gv, int vi, int vj, int vk, unsigned centre, int dx, int dy, int dz) {
  return lut_entry(gv, vi, vj, vk, centre, dx, dy, dz);
}

// KDTREE: record index of voxel (i+dx, ...) if it is in the centroid cloud (valid or rejected) and
// its f32 centroid is closer than the radius -- radiusSearch, voxel_grid_covariance_omp.h:476-505,
// [FLANN] L2_Simple accumulated in f32, RadiusResultSet keeps dist < r^2.  Else -1.
__device__ __forceinline__ int probe_kd(const GridView& gv, int vi, int vj, int vk, unsigned centre, int dx, int dy, int dz, float x,
                                        float y, float z, float r2) {
  const int e = lut_entry(gv, vi, vj, vk, centre, dx, dy, dz);
  if (e == kLutEmpty) return -1;
  const int rix = (e >= 0) ? e : -(e + 2);
  const float4 c = *reinterpret_cast<const float4*>(gv.centroids + rix);  // (cx, cy, cz, -)
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

struct RecRegs {
  double mx, my, mz;
  f2 p0, p1, p2, p3;  // (c00,c01) (c01,c11) (c02,c12) (c11,c22)
};
// REC4: a fourth load for the (c01,c11) pair instead of assembling it (see VoxelRec)
template <bool REC4 = false>
__device__ __forceinline__ RecRegs load_rec(const VoxelRec* __restrict__ recs, int r) {
  const float4* p = reinterpret_cast<const float4*>(recs + r);
  const float4 a = p[0], b = p[1], c = p[2];  // mean x, y | mean z, c00, c01 | c02, c12, c11, c22
  RecRegs o;
  o.mx = __hiloint2double(__float_as_int(a.y), __float_as_int(a.x));
  o.my = __hiloint2double(__float_as_int(a.w), __float_as_int(a.z));
  o.mz = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x));
  o.p0 = f2{b.z, b.w};
  if (REC4) {
    const float2 d = reinterpret_cast<const float2*>(recs + r)[7];
    o.p1 = f2{d.x, d.y};
  } else {
    o.p1 = f2{b.w, c.z};
  }
  o.p2 = f2{c.x, c.y};
  o.p3 = f2{c.z, c.w};
  return o;
}

// Per-evaluation coefficient tables of computePointDerivatives (ndt_omp_impl.hpp:398-440) in the pair layout the
// packed math consumes.  With v = (x, y, z) the point, row r of j_ang / h_ang gives  j_r = j_ang[r] . v,
// h_r = h_ang[r] . v; pair (r, r') of the table holds (M[r][c], M[r'][c]) for c = 0, 1, 2:
//   jp: (j2,j5) (j3,j6) (j4,j7) (j0,j1)                       -- columns 1, 2 of B = dR/d(angle) x, then column 0
//   hp: (h0,h2) (h1,h3) (h6,h9) (h7,h10) (h8,h11) (h4,h5) (h12,h13) ; h14 on its own
struct PackedTables {
  f2 jp[3][4];
  f2 hp[3][7];
  float h14[3];
  float pad_;
};
static_assert(sizeof(PackedTables) == 70 * sizeof(float), "PackedTables is addressed as 70 floats");
// float index inside PackedTables of element e of the row tables (e < 24: j[e / 3][e % 3]; else h[(e - 24) / 3][(e - 24) % 3])
__device__ __constant__ unsigned char kPackPos[69] = {6, 14, 22, 7, 15, 23, 0, 8, 16, 2, 10, 18, 4, 12, 20, 1, 9, 17, 3, 11, 19, 5, 13, 21, 24, 38, 52, 26, 40, 54, 25, 39, 53, 27, 41, 55, 34, 48, 62, 35, 49, 63, 28, 42, 56, 30, 44, 58, 32, 46, 60, 29, 43, 57, 31, 45, 59, 33, 47, 61, 36, 50, 64, 37, 51, 65, 66, 67, 68};
// EvalParams (row tables j[8][3], h[15][3]) -> PackedTables, one element per thread
__device__ __forceinline__ void pack_tables(const EvalParams& P, PackedTables& t, int tid, int nthreads) {
  const float* src = &P.j[0][0];  // j[8][3] and h[15][3] are contiguous: 69 floats
  float* dst = reinterpret_cast<float*>(&t);
  for (int e = tid; e < 69; e += nthreads) dst[kPackPos[e]] = src[e];
}

// per-point pieces of computePointDerivatives (f32), in pairs
struct PointDeriv {
  f2 jp[4];   // (j2,j5) (j3,j6) (j4,j7) (j0,j1)
  f2 hp[7];   // (h0,h2) (h1,h3) (h6,h9) (h7,h10) (h8,h11) (h4,h5) (h12,h13)
  float h14;
};
template <bool WANT_H>
__device__ __forceinline__ void point_derivatives(const PackedTables& t, float x, float y, float z, PointDeriv& d) {
  const f2 xy = f2{x, y}, zz = f2{z, z};
#pragma unroll
  for (int q = 0; q < 4; q++) d.jp[q] = pk_fma_b0lo(zz, t.jp[2][q], pk_fma_b0hi(xy, t.jp[1][q], pk_mul_b0lo(xy, t.jp[0][q])));
  if (WANT_H) {
#pragma unroll
    for (int q = 0; q < 7; q++) d.hp[q] = pk_fma_b0lo(zz, t.hp[2][q], pk_fma_b0hi(xy, t.hp[1][q], pk_mul_b0lo(xy, t.hp[0][q])));
    d.h14 = fma1(z, t.h14[2], fma1(y, t.h14[1], mul1(x, t.h14[0])));
  }
}

// ---------------------------------------------------------------------------
// updateDerivatives (ndt_omp_impl.hpp:484-537) in factored form.  With xc = C x' (C symmetric) the
// reference's per-neighbour quantities are
//   gradient_k     += e * (xc . J_k)
//   hessian(i,j)   += e * ( -d2 (xc . J_i)(xc . J_j) + xc . HE_ij + J_j^T C J_i )
// and J_k, HE_ij depend on the POINT only.  Everything is therefore linear in two small
// per-neighbour objects,
//   xe = sum_n e_n xc_n                      (3)
//   A  = sum_n e_n (C_n - d2 xc_n xc_n^T)    (3x3 symmetric, 6)
// which are accumulated over the <= 7 neighbours of a point (f32), after which
//   gradient = J^T xe ,  hessian = J^T A J + [xe . HE_ij]   are formed ONCE per point and added to
// the f64 accumulators.  The f32 rounding of the per-point sums differs from the reference's
// per-neighbour rounding by O(1e-7) relative per point (same order as its own f32 noise), far
// inside the parity tolerance; e itself follows the reference's f32 / f64 cast sequence (:499-510).
// acc: [0]=score [1..6]=gradient [7..27]=Hessian upper triangle [28]=neighbour count
// ---------------------------------------------------------------------------
struct PointAcc {
  f2 xe01;    // (xe0, xe1)
  f2 xe2_;    // (xe2, -)
  f2 a0001;   // (a00, a01)
  f2 a0212;   // (a02, a12)
  f2 a1122;   // (a11, a22)
  double score;
  int nn;
};
__device__ __forceinline__ PointAcc point_acc_zero() {
  PointAcc pa;
  pa.xe01 = f2{0.f, 0.f};
  pa.xe2_ = f2{0.f, 0.f};
  pa.a0001 = f2{0.f, 0.f};
  pa.a0212 = f2{0.f, 0.f};
  pa.a1122 = f2{0.f, 0.f};
  pa.score = 0.0;
  pa.nn = 0;
  return pa;
}

template <bool WANT_H>
__device__ __forceinline__ void accumulate_neighbor_factored(PointAcc& pa, float x0, float x1, float x2, const RecRegs& r,
                                                             double d1, float d2) {
  const f2 x01 = f2{x0, x1}, x2b = f2{x2, x2};
  // (xc0, xc1) = x0 (c00,c01) + x1 (c01,c11) + x2 (c02,c12) ;  xc2 = x0 c02 + x1 c12 + x2 c22
  const f2 xc01 = pk_fma_b0lo(x2b, r.p2, pk_fma_b0hi(x01, r.p1, pk_mul_b0lo(x01, r.p0)));
  const f2 u = pk_mul(x01, r.p2);
  const float xc2 = fma1(x2, r.p3.y, add1(u.x, u.y));
  const f2 v = pk_mul(x01, xc01);
  const float q = fma1(x2, xc2, add1(v.x, v.y));
  float e = expf(mul1(mul1(-d2, q), 0.5f));                                  // :499
  const float score_inc = static_cast<float>(-d1 * static_cast<double>(e));  // :501
  e = mul1(d2, e);                                                           // :503
  if (e > 1.0f || e < 0.0f || e != e) return;                                // :506-507 (adds nothing, not even the score)
  e = static_cast<float>(static_cast<double>(e) * d1);                       // :510
  pa.score += static_cast<double>(score_inc);
  pa.nn += 1;
  const f2 em = f2{e, mul1(-d2, e)};  // (e, -d2 e)
  pa.xe01 = pk_fma_b0lo(em, xc01, pa.xe01);
  pa.xe2_.x = fma1(e, xc2, pa.xe2_.x);
  if (WANT_H) {
    const f2 t01 = pk_mul_b0hi(em, xc01);  // -d2 e (xc0, xc1)
    const float t2 = mul1(em.y, xc2);
    const f2 xc2b = f2{xc2, xc2};
    pa.a0001 = pk_fma_b0lo(t01, xc01, pk_fma_b0lo(em, r.p0, pa.a0001));  // += e (c00,c01) + t0 (xc0,xc1)
    pa.a0212 = pk_fma(t01, xc2b, pk_fma_b0lo(em, r.p2, pa.a0212));       // += e (c02,c12) + (t0,t1) xc2
    pa.a1122.x = fma1(t01.y, xc01.y, fma1(e, r.p3.x, pa.a1122.x));       // += e c11 + t1 xc1
    pa.a1122.y = fma1(t2, xc2, fma1(e, r.p3.y, pa.a1122.y));             // += e c22 + t2 xc2
  }
}

// J_E = [ I3 | B ],  B columns: (0, j0, j1), (j2, j3, j4), (j5, j6, j7)   (ndt_omp_impl.hpp:407-414)
template <bool WANT_H>
__device__ __forceinline__ void finish_point(double (&acc)[kNumAcc], const PointAcc& pa, const PointDeriv& d) {
  const f2 xe01 = pa.xe01, xe2 = pa.xe2_;
  const f2 jp0 = d.jp[0], jp1 = d.jp[1], jp2 = d.jp[2], jp3 = d.jp[3];
  acc[0] += pa.score;
  acc[28] += static_cast<double>(pa.nn);
  acc[1] += static_cast<double>(xe01.x);
  acc[2] += static_cast<double>(xe01.y);
  acc[3] += static_cast<double>(xe2.x);
  // g3 = xe1 j0 + xe2 j1 ; (g4, g5) = xe0 (j2,j5) + xe1 (j3,j6) + xe2 (j4,j7)
  acc[4] += static_cast<double>(fma1(xe2.x, jp3.y, mul1(xe01.y, jp3.x)));
  const f2 g45 = pk_fma_b0lo(xe2, jp2, pk_fma_b0hi(xe01, jp1, pk_mul_b0lo(xe01, jp0)));
  acc[5] += static_cast<double>(g45.x);
  acc[6] += static_cast<double>(g45.y);
  if (WANT_H) {
    const f2 a0001 = pa.a0001, a0212 = pa.a0212, a1122 = pa.a1122;
    // AB = A B.  Columns 1, 2 as pairs, row r:  A[r][0] (j2,j5) + A[r][1] (j3,j6) + A[r][2] (j4,j7)
    const f2 ab0 = pk_fma_b0lo(a0212, jp2, pk_fma_b0hi(a0001, jp1, pk_mul_b0lo(a0001, jp0)));  // A[0] = (a00, a01, a02)
    const f2 ab1 = pk_fma_b0hi(a0212, jp2, pk_fma_b0lo(a1122, jp1, pk_mul_b0hi(a0001, jp0)));  // A[1] = (a01, a11, a12)
    const f2 ab2 = pk_fma_b0hi(a1122, jp2, pk_fma_b0hi(a0212, jp1, pk_mul_b0lo(a0212, jp0)));  // A[2] = (a02, a12, a22)
    // column 0:  A[r][1] j0 + A[r][2] j1
    const float ab00 = fma1(a0212.x, jp3.y, mul1(a0001.y, jp3.x));
    const float ab10 = fma1(a0212.y, jp3.y, mul1(a1122.x, jp3.x));
    const float ab20 = fma1(a1122.y, jp3.y, mul1(a0212.y, jp3.x));
    // x-block of H_E: a b c / b d e / c e f  with a=(0,h0,h1) b=(0,h2,h3) c=(0,h4,h5) d=(h6,h7,h8) e=(h9,h10,h11) f=(h12,h13,h14)
    const f2 xab = pk_fma_b0lo(xe2, d.hp[1], pk_mul_b0hi(xe01, d.hp[0]));                               // (xa, xb)
    const f2 xde = pk_fma_b0lo(xe2, d.hp[4], pk_fma_b0hi(xe01, d.hp[3], pk_mul_b0lo(xe01, d.hp[2])));  // (xd, xe)
    const float xcc = fma1(xe2.x, d.hp[5].y, mul1(xe01.y, d.hp[5].x));
    const f2 w = pk_mul(xe01, d.hp[6]);
    const float xf = fma1(xe2.x, d.h14, add1(w.x, w.y));
    // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
    acc[7] += static_cast<double>(a0001.x);
    acc[8] += static_cast<double>(a0001.y);
    acc[9] += static_cast<double>(a0212.x);
    acc[10] += static_cast<double>(ab00);
    acc[11] += static_cast<double>(ab0.x);
    acc[12] += static_cast<double>(ab0.y);
    acc[13] += static_cast<double>(a1122.x);
    acc[14] += static_cast<double>(a0212.y);
    acc[15] += static_cast<double>(ab10);
    acc[16] += static_cast<double>(ab1.x);
    acc[17] += static_cast<double>(ab1.y);
    acc[18] += static_cast<double>(a1122.y);
    acc[19] += static_cast<double>(ab20);
    acc[20] += static_cast<double>(ab2.x);
    acc[21] += static_cast<double>(ab2.y);
    // B^T (A B) + X ;  B[:,0] = (0, j0, j1)  B[:,1] = (j2, j3, j4)  B[:,2] = (j5, j6, j7)
    const float v00 = fma1(jp3.y, ab20, mul1(jp3.x, ab10));
    const f2 v0x = pk_fma_b0hi(jp3, ab2, pk_mul_b0lo(jp3, ab1));                              // (v01, v02)
    const f2 v1x = pk_fma_b0lo(jp2, ab2, pk_fma_b0lo(jp1, ab1, pk_mul_b0lo(jp0, ab0)));      // (v11, v12)
    const float v22 = fma1(jp2.y, ab2.y, fma1(jp1.y, ab1.y, mul1(jp0.y, ab0.y)));
    acc[22] += static_cast<double>(add1(v00, xab.x));
    acc[23] += static_cast<double>(add1(v0x.x, xab.y));
    acc[24] += static_cast<double>(add1(v0x.y, xcc));
    const f2 h45 = pk_add(v1x, xde);
    acc[25] += static_cast<double>(h45.x);
    acc[26] += static_cast<double>(h45.y);
    acc[27] += static_cast<double>(add1(v22, xf));
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
#ifdef NDT_THROUGHPUT_UNIT
constexpr bool kLimitRecordLoads = true;
#else
constexpr bool kLimitRecordLoads = false;
#endif
// LIMIT: at most two voxel records in flight per point (see the neighbour loop): the throughput kernels and the one-launch
// kernel of the launch path, whose blocks share CUs with other blocks; not the evaluation server (one block per CU).
template <int NNB, bool WANT_H, bool STAMP = false, bool PRELOADED = false, bool LIMIT = kLimitRecordLoads, bool REC4 = false>
__device__ __forceinline__ void derivatives_body(const float4* __restrict__ src, int n, const GridView& gv, const EvalParams& prm,
                                                 const PackedTables& tab, int first, int stride, double (&acc)[kNumAcc],
                                                 unsigned long long* st = nullptr, float4 first_pt = float4{0.f, 0.f, 0.f, 0.f}) {
  for (int i = first; i < n; i += stride) {
    const float4 pt = (PRELOADED && i == first) ? first_pt : src[i];
    if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[1] = stamp(); }
    float tx, ty, tz;
    xform_point(prm.T, pt.x, pt.y, pt.z, tx, ty, tz);
    // A non-finite point has no neighbourhood in the reference (its voxel index is garbage and fails
    // the bounding-box test): it contributes nothing.  Without this, 0 x NaN of its point derivatives
    // would poison the per-point finish even though every neighbour term is rejected.
    int vi, vj, vk;
    search_ijk(gv.g, tx, ty, tz, vi, vj, vk);
    // ONE structured region per point (no early `continue`s): the 29 f64 accumulators are then updated in place --
    // with several exits the compiler kept two register sets for them and copied all of them twice per point
    if (finite3(tx, ty, tz) && near_grid(gv.g, vi, vj, vk)) {
      const unsigned centre = lut_index(gv.g, vi, vj, vk);
      int rec[NNB];
      bool any = false;
      if (NNB == 7 && gv.g.hash_bits == 0) {
        // the voxel and its two x neighbours are three consecutive table entries: one 12-byte load instead of three probes
        // (every load instruction of every wave goes through the CU's one vector memory path); same entries, same order
        struct __attribute__((packed, aligned(4))) Int3 { int a, b, c; };
        const Int3 t = *reinterpret_cast<const Int3*>(gv.lut + (centre - 1u));
        rec[0] = t.b;
        rec[1 % NNB] = t.c;
        rec[2 % NNB] = t.a;
#pragma unroll
        for (int k = 3; k < NNB; k++) {
          int dx, dy, dz;
          nb_offset<NNB>(k, dx, dy, dz);
          rec[k] = probe(gv, vi, vj, vk, centre, dx, dy, dz);
        }
#pragma unroll
        for (int k = 0; k < NNB; k++) any |= (rec[k] >= 0);
      } else {
#pragma unroll
        for (int k = 0; k < NNB; k++) {
          int dx, dy, dz;
          nb_offset<NNB>(k, dx, dy, dz);
          rec[k] = probe(gv, vi, vj, vk, centre, dx, dy, dz);
          any |= (rec[k] >= 0);
        }
      }
      if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[2] = stamp(); }
      if (any) {
        const double txd = static_cast<double>(tx), tyd = static_cast<double>(ty), tzd = static_cast<double>(tz);
        // Software pipeline over the neighbours: the record of neighbour k+1 is requested (index
        // clamped, so the load is unconditional and hoistable) before neighbour k's math runs; one
        // record gather latency is exposed per point instead of one per neighbour.
        PointAcc pa = point_acc_zero();
        RecRegs cur = load_rec<REC4>(gv.recs, rec[0] < 0 ? 0 : rec[0]);
        if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[3] = stamp(); }
#pragma unroll
        for (int k = 0; k < NNB; k++) {
          RecRegs nxt = cur;
          // LIMIT: keep the compiler from hoisting all seven record loads to the top of the point -- with two records in
          // flight instead of seven the with-Hessian kernel needs 126 VGPRs instead of 207, four waves per SIMD instead of
          // two, and a 512-scan map build runs 9-13 % faster.  The evaluation server (one block per CU by design) keeps
          // every load in flight: limited, a 2M-point registration through it is 11 % slower.
          if (LIMIT && WANT_H) asm volatile("" ::: "memory");
          if (k + 1 < NNB) nxt = load_rec<REC4>(gv.recs, rec[k + 1] < 0 ? 0 : rec[k + 1]);
          if (rec[k] >= 0) {
            // x_trans (f32 -> f64) - mean (f64), rounded to f32  (:259-262, :492)
            const float x0 = static_cast<float>(txd - cur.mx);
            const float x1 = static_cast<float>(tyd - cur.my);
            const float x2 = static_cast<float>(tzd - cur.mz);
            accumulate_neighbor_factored<WANT_H>(pa, x0, x1, x2, cur, prm.d1, prm.d2);
          }
          cur = nxt;
        }
        // the point's own derivative pairs only now: 23 registers that the neighbour loop does not have to carry
        PointDeriv d;
        point_derivatives<WANT_H>(tab, pt.x, pt.y, pt.z, d);
        finish_point<WANT_H>(acc, pa, d);
      }
    }
    if (STAMP) st[4] = stamp();
  }
}

// KDTREE search: 3x3x3 cells around the point, centroid-distance filter, same factored math.
// (The reference visits the hits sorted by distance; only the f64 summation order differs.)
template <bool WANT_H>
__device__ __forceinline__ void derivatives_body_kd(const float4* __restrict__ src, int n, const GridView& gv, const EvalParams& prm,
                                                    const PackedTables& tab, int first, int stride, double (&acc)[kNumAcc]) {
  const float r2 = __int_as_float(prm.pad);
  for (int i = first; i < n; i += stride) {
    const float4 pt = src[i];
    float tx, ty, tz;
    xform_point(prm.T, pt.x, pt.y, pt.z, tx, ty, tz);
    // non-finite point: no neighbourhood in the reference (see derivatives_body)
    if (!finite3(tx, ty, tz)) continue;
    int vi, vj, vk;
    search_ijk(gv.g, tx, ty, tz, vi, vj, vk);
    if (!near_grid(gv.g, vi, vj, vk)) continue;
    const unsigned centre = lut_index(gv.g, vi, vj, vk);
    PointAcc pa = point_acc_zero();
    for (int a = -1; a <= 1; a++)
      for (int b = -1; b <= 1; b++)
        for (int c = -1; c <= 1; c++) {
          if (kLimitRecordLoads) asm volatile("" ::: "memory");  // one cell's probe / centroid / record loads at a time (VGPRs)
          const int rix = probe_kd(gv, vi, vj, vk, centre, a, b, c, tx, ty, tz, r2);
          if (rix < 0) continue;
          const RecRegs r = load_rec(gv.recs, rix);
          const float x0 = static_cast<float>(static_cast<double>(tx) - r.mx);
          const float x1 = static_cast<float>(static_cast<double>(ty) - r.my);
          const float x2 = static_cast<float>(static_cast<double>(tz) - r.mz);
          accumulate_neighbor_factored<WANT_H>(pa, x0, x1, x2, r, prm.d1, prm.d2);
        }
    if (pa.nn != 0) {
      PointDeriv d;
      point_derivatives<WANT_H>(tab, pt.x, pt.y, pt.z, d);
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
    const unsigned centre = lut_index(gv.g, vi, vj, vk);
    for (int k = 0; k < NNB; k++) {
      int dx, dy, dz;
      nb_offset<NNB>(k, dx, dy, dz);
      // (throughput unit: one neighbour's nine f64 words at a time -- hoisted to the top of the point, the seven
      // neighbours' loads alone are 126 VGPRs, and the mixed batch kernel that contains this body is sized by it)
      if (kLimitRecordLoads) asm volatile("" ::: "memory");
      const int rix = (NNB == 27) ? probe_kd(gv, vi, vj, vk, centre, dx, dy, dz, tx, ty, tz, static_cast<float>(prm.r2))
                                  : probe(gv, vi, vj, vk, centre, dx, dy, dz);
      if (rix < 0) continue;
      // mean from the record, the inverse covariance in f64 from the record's side sector (the leaf's icov_, :592)
      const double* mu = gv.recs[rix].mean;
      const double* ic = gv.centroids[rix].icov;
      const double mx = mu[0], my = mu[1], mz = mu[2];
      const double c00 = ic[0], c01 = ic[1], c02 = ic[2], c11 = ic[3], c12 = ic[4], c22 = ic[5];
      const double x0 = static_cast<double>(tx) - mx, x1 = static_cast<double>(ty) - my, x2 = static_cast<double>(tz) - mz;
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
