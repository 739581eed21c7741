// ndt_kernels.hip -- hand-written HIP kernels of the MI355X NDT core (gfx950, wave64): the throughput side of the
// evaluation.  (K1, everything that builds voxel structures, is ndt_grid_kernels.hip / ndt_sparse.hip; the single-scan
// latency path -- one-launch evaluation and the persistent evaluation server -- is ndt_latency.hip; all share ndt_device.hpp.)
//
//  K2  per-evaluation score / gradient / Hessian (computeDerivatives, ndt_omp_impl.hpp:179-285 with updateDerivatives
//      :484-537 fused with the f32 point transform): k_derivatives (one launch per evaluation, also over a whole lock-step
//      batch), k_batch_step (mixed-kind batch steps), the all-f64 Hessian k_hessian64 (computeHessian :540-645), k_reduce
//      (fixed-order sum of the per-block rows), calculateScore (:935-983), getFitnessScore (k_fitness) and the search index
//      it walks (k_cell_ranges, k_gather_points).
//
// Gather work: no MFMA (there is no dense contraction).  Loads are 16 B per lane (float4 points, 3 x dwordx4 per 64-B
// voxel record), the LUT probe + record gather is served from L2 / Infinity Cache for the target sizes of interest, and the
// 29 f64 accumulators are reduced with a VALU-only wave64 fold, then LDS across the waves of a block, then a fixed-order
// sum over the blocks.
#define NDT_THROUGHPUT_UNIT 1  // see derivatives_body: records in flight per point
#include "ndt_device.hpp"
#include "ndt_search.hpp"

namespace ndt {

namespace {

// test hook: every thread contributes acc[k] = f(global thread, k); out[block][32]
__global__ __launch_bounds__(kBlock) void k_selftest_reduce(double* __restrict__ out) {
  __shared__ double lds[(kBlock / kWave) * 32];
  double acc[kNumAcc];
  const int gt = blockIdx.x * kBlock + threadIdx.x;
#pragma unroll
  for (int k = 0; k < kNumAcc; k++) acc[k] = 0.5 * static_cast<double>((gt * 131 + k * 17 + (gt >> 3) * k) % 1009) - 100.0;
  block_reduce_store<kNumAcc>(acc, out + static_cast<size_t>(blockIdx.x) * kEvalStride, lds);
}


// Diagnostic build of the DIRECT7 derivative kernel with s_memtime stamps (never used by the
// product path): per wave, cycles at entry / point arrived / LUT arrived / first record arrived /
// neighbour math done / wave fold done / block done.
__global__ __launch_bounds__(kBlock) void k_derivatives_stamped(const float4* __restrict__ src, int n, GridView gv,
                                                                EvalParams P, double* __restrict__ partials,
                                                                unsigned long long* __restrict__ stamps) {
  __shared__ double lds[(kBlock / kWave) * 32];
  __shared__ PackedTables sT;
  pack_tables(P, sT, threadIdx.x, kBlock);
  __syncthreads();
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  st[0] = stamp();
  double acc[kNumAcc];
#pragma unroll
  for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
  const int first = blockIdx.x * kBlock + threadIdx.x, stride = gridDim.x * kBlock;
  derivatives_body<7, true, true>(src, n, gv, P, sT, first, stride, acc, st);
  const double tot = wave_fold<kNumAcc>(acc);
  st[5] = stamp();
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if ((lane & 1) == 0) lds[wave * 32 + fold_index(lane)] = tot;
  __syncthreads();
  if (threadIdx.x < kNumAcc) {
    double v = lds[threadIdx.x];
    for (int w = 1; w < kBlock / kWave; w++) v += lds[w * 32 + threadIdx.x];
    partials[static_cast<size_t>(blockIdx.x) * kEvalStride + threadIdx.x] = v;
  }
  st[6] = stamp();
  if (lane == 0) {
    unsigned long long* o = stamps + (static_cast<size_t>(blockIdx.x) * (kBlock / kWave) + wave) * 8;
    for (int k = 0; k < 8; k++) o[k] = st[k];
  }
}

template <int NNB, bool WANT_H, bool BATCH>
__global__ __launch_bounds__(kBlock) void k_derivatives(const float4* __restrict__ src, int n, GridView gv, EvalParams P,
                                                        const ScanDesc* __restrict__ descs, const int* __restrict__ active,
                                                        int max_blocks, double* __restrict__ partials) {
  __shared__ double lds[(kBlock / kWave) * 32];
  __shared__ EvalParams sP;
  __shared__ PackedTables sT;
  double acc[kNumAcc];
#pragma unroll
  for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
  const int first = blockIdx.x * kBlock + threadIdx.x;
  // parameters through LDS: as kernel arguments they overflow the SGPR file.  BATCH: grid.y walks the scans that asked
  // for THIS kind of evaluation in this step
  const int scan = BATCH ? active[blockIdx.y] : 0;
  const ScanDesc* dsc = BATCH ? descs + scan : nullptr;
  {
    const int* sp = BATCH ? reinterpret_cast<const int*>(&dsc->P) : reinterpret_cast<const int*>(&P);
    int* dp = reinterpret_cast<int*>(&sP);
    for (int t = threadIdx.x; t < static_cast<int>(sizeof(EvalParams) / 4); t += kBlock) dp[t] = sp[t];
    pack_tables(*reinterpret_cast<const EvalParams*>(sp), sT, threadIdx.x, kBlock);
  }
  __syncthreads();
  const float4* pts = BATCH ? src + dsc->offset : src;
  int cnt = BATCH ? dsc->count : n;
  // BATCH: a scan is walked by ITS OWN number of blocks (descs[scan].pad, a function of its size only), whatever the
  // grid is: its sums do not depend on which other scans share the launch.  Block b takes one CONTIGUOUS run of
  // kBatchPointsPerBlock points of the (spatially ordered) scan, the runs dealt to the XCDs in eighths (xcd_chunk): every
  // XCD's L2 then works on one compact region of the voxel records instead of all eight caching the whole map.
  int stride = static_cast<int>(gridDim.x) * kBlock, first_pt = first;
  if (BATCH) {
    if (static_cast<int>(blockIdx.x) >= dsc->pad) return;
    const int lo = xcd_chunk(blockIdx.x, dsc->pad) * kBatchPointsPerBlock;
    first_pt = lo + static_cast<int>(threadIdx.x);
    cnt = min(cnt, lo + kBatchPointsPerBlock);
    stride = kBlock;
  }
  if (NNB == 27) derivatives_body_kd<WANT_H>(pts, cnt, gv, sP, sT, first_pt, stride, acc);
  else derivatives_body<NNB == 27 ? 7 : NNB, WANT_H>(pts, cnt, gv, sP, sT, first_pt, stride, acc);
  block_reduce_store<kNumAcc>(acc, partials + (static_cast<size_t>(scan) * max_blocks + blockIdx.x) * kEvalStride, lds);
}

template <int NNB, bool BATCH>
__global__ __launch_bounds__(kBlock) void k_hessian64(const float4* __restrict__ src, int n, GridView gv, Hess64Params P,
                                                      const ScanDesc* __restrict__ descs, const int* __restrict__ active,
                                                      int max_blocks, double* __restrict__ partials) {
  __shared__ double lds[(kBlock / kWave) * 32];
  __shared__ Hess64Params sP;
  double acc[kNumAcc];
  for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
  const Hess64Params* prm = &P;
  int scan = 0;
  if (BATCH) {
    scan = active[blockIdx.y];
    const ScanDesc* dsc = descs + scan;
    const int* sp = reinterpret_cast<const int*>(&dsc->P64);
    int* dp = reinterpret_cast<int*>(&sP);
    for (int t = threadIdx.x; t < static_cast<int>(sizeof(Hess64Params) / 4); t += kBlock) dp[t] = sp[t];
    __syncthreads();
    src += dsc->offset;
    n = dsc->count;
    prm = &sP;
  } else {
    const int* sp = reinterpret_cast<const int*>(&P);
    int* dp = reinterpret_cast<int*>(&sP);
    for (int t = threadIdx.x; t < static_cast<int>(sizeof(Hess64Params) / 4); t += kBlock) dp[t] = sp[t];
    __syncthreads();
    prm = &sP;
  }
  double* out = partials + (static_cast<size_t>(scan) * max_blocks + blockIdx.x) * kEvalStride;
  if (BATCH) {  // a scan's own block count, one contiguous run of points per block (see k_derivatives)
    const int nb = descs[scan].pad;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    const int lo = xcd_chunk(blockIdx.x, nb) * kBatchPointsPerBlock;
    hessian64_body<NNB>(src, min(n, lo + kBatchPointsPerBlock), gv, *prm, lo + static_cast<int>(threadIdx.x), kBlock, acc);
  } else {
    hessian64_body<NNB>(src, n, gv, *prm, blockIdx.x * kBlock + threadIdx.x, static_cast<int>(gridDim.x) * kBlock, acc);
  }
  block_reduce_store<kNumAcc>(acc, out, lds);
}

// One launch for a lock-step batch step in which the scans ask for DIFFERENT kinds of evaluation
// (late in a batch: a few scans still in their line search, some recomputing the f64 Hessian).  (Measured: giving the f64
// scans a launch of their own takes this kernel from 152 to 126 VGPRs -- four waves per SIMD instead of three -- but the
// extra ~28 us launch in most tail steps costs more than that buys: 6.05k instead of 6.4k reg/s on the 512-scan build.)
// grid.y walks all live scans; the kind is block-uniform (read from the scan's descriptor).  Three
// separate launches of the specialised kernels serialise and each pays its own ramp-up; steps in which
// every live scan wants the same kind keep using those.
template <int NNB>
__global__ __launch_bounds__(kBlock) void k_batch_step(const float4* __restrict__ src, GridView gv,
                                                       const ScanDesc* __restrict__ descs, const int* __restrict__ active,
                                                       int max_blocks, double* __restrict__ partials) {
  __shared__ double lds[(kBlock / kWave) * 32];
  __shared__ EvalParams sP;
  __shared__ Hess64Params sP64;
  __shared__ PackedTables sT;
  const int scan = active[blockIdx.y];
  const ScanDesc* dsc = descs + scan;
  const int kind = dsc->kind;
  {
    const int* sp = (kind == 2) ? reinterpret_cast<const int*>(&dsc->P64) : reinterpret_cast<const int*>(&dsc->P);
    int* dp = (kind == 2) ? reinterpret_cast<int*>(&sP64) : reinterpret_cast<int*>(&sP);
    const int words = static_cast<int>(((kind == 2) ? sizeof(Hess64Params) : sizeof(EvalParams)) / 4);
    for (int t = threadIdx.x; t < words; t += kBlock) dp[t] = sp[t];
    if (kind != 2) pack_tables(dsc->P, sT, threadIdx.x, kBlock);
  }
  __syncthreads();
  double acc[kNumAcc];
#pragma unroll
  for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
  if (static_cast<int>(blockIdx.x) >= dsc->pad) return;  // (uniform; after the barriers above)
  const int lo = xcd_chunk(blockIdx.x, dsc->pad) * kBatchPointsPerBlock;  // one contiguous run of points per block (see k_derivatives)
  const int first = lo + static_cast<int>(threadIdx.x), stride = kBlock;
  const float4* pts = src + dsc->offset;
  const int n = min(dsc->count, lo + kBatchPointsPerBlock);
  if (kind == 2) {
    hessian64_body<NNB, true>(pts, n, gv, sP64, first, stride, acc);
  } else if (NNB == 27) {
    if (kind == 0) derivatives_body_kd<true>(pts, n, gv, sP, sT, first, stride, acc);
    else derivatives_body_kd<false>(pts, n, gv, sP, sT, first, stride, acc);
  } else {
    if (kind == 0) derivatives_body<NNB == 27 ? 7 : NNB, true>(pts, n, gv, sP, sT, first, stride, acc);
    else derivatives_body<NNB == 27 ? 7 : NNB, false>(pts, n, gv, sP, sT, first, stride, acc);
  }
  block_reduce_store<kNumAcc>(acc, partials + (static_cast<size_t>(scan) * max_blocks + blockIdx.x) * kEvalStride, lds);
}

// ---------------------------------------------------------------------------
// fixed-order reduction of the per-block partials
// ---------------------------------------------------------------------------
constexpr int kReduceThreads = 1024;
// n_blocks: rows per scan (single scan) or row stride per scan (batch, where descs[scan].pad holds
// the number of rows actually written this step)
__global__ __launch_bounds__(kReduceThreads) void k_reduce(const double* __restrict__ partials, int n_blocks,
                                                           const ScanDesc* __restrict__ descs, double* __restrict__ out,
                                                           unsigned long long seq) {
  const int scan = blockIdx.x;
  int rows = n_blocks;
  if (descs) {
    if (descs[scan].kind == 3) return;  // EVAL_NONE: row left untouched
    rows = descs[scan].pad;
  }
  constexpr int kParts = kReduceThreads / kEvalStride;  // 32
  const int k = threadIdx.x % kEvalStride, part = threadIdx.x / kEvalStride;
  const double* base = partials + static_cast<size_t>(scan) * n_blocks * kEvalStride;
  double v = 0.0;
  if (k < kNumAcc) {
    int b = part;
    for (; b + 3 * kParts < rows; b += 4 * kParts) {  // 4 independent loads in flight
      const double a0 = base[static_cast<size_t>(b) * kEvalStride + k];
      const double a1 = base[static_cast<size_t>(b + kParts) * kEvalStride + k];
      const double a2 = base[static_cast<size_t>(b + 2 * kParts) * kEvalStride + k];
      const double a3 = base[static_cast<size_t>(b + 3 * kParts) * kEvalStride + k];
      v += a0; v += a1; v += a2; v += a3;
    }
    for (; b < rows; b += kParts) v += base[static_cast<size_t>(b) * kEvalStride + k];
  }
  __shared__ double s[kParts][kEvalStride];
  s[part][k] = v;
  __syncthreads();
  if (threadIdx.x < kEvalStride) {
    double t = 0.0;
#pragma unroll
    for (int p = 0; p < kParts; p++) t += s[p][threadIdx.x];
    // slot 31 is the completion word when the row is polled from the host (seq != 0)
    if (seq == 0) out[static_cast<size_t>(scan) * kEvalStride + threadIdx.x] = t;
    else publish_row(out + static_cast<size_t>(scan) * kEvalStride, t, seq);
  }
}

// Rows of the exchange buffer (after the all-reduce of a sharded lock-step) -> the pinned host block the host polls:
// every row, then its sequence word in slot 31 (publish_row).
__global__ __launch_bounds__(kWave) void k_publish_rows(const double* __restrict__ rows, double* __restrict__ out,
                                                       unsigned long long seq) {
  const size_t r = blockIdx.x;
  const double v = (threadIdx.x < kEvalStride - 1) ? rows[r * kEvalStride + threadIdx.x] : 0.0;
  publish_row(out + r * kEvalStride, v, seq);
}

// transformed source cloud ("output" of align), w = 1.  dense = 0: [PCL] transformPointCloud leaves
// non-finite points as they are.
__global__ __launch_bounds__(kBlock) void k_transform(const float4* __restrict__ src, int n, EvalParams P,
                                                      float4* __restrict__ dst, int dense) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 pt = src[i];
    float tx = pt.x, ty = pt.y, tz = pt.z;
    if (dense || finite3(pt.x, pt.y, pt.z)) xform_point(P.T, pt.x, pt.y, pt.z, tx, ty, tz);
    dst[i] = make_float4(tx, ty, tz, 1.0f);
  }
}

// ---------------------------------------------------------------------------
// N4: [PCL 1.10] Registration::getFitnessScore -- mean squared distance from every transformed source
// point to its nearest target point.  Exact nearest neighbour over the target's own voxel grid (K1's
// counting sort already groups the target points by cell): scan the query's cell, then cubic shells
// of cells around it, and stop once the best distance cannot be beaten by any unvisited shell.
// Distances as [FLANN] L2_Simple computes them: f32, (dx*dx + dy*dy) + dz*dz, no contraction.
// acc[0] = sum of the accepted squared distances (f64), acc[1] = their count.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cell_ranges(const int* __restrict__ leaf_cell, const unsigned* __restrict__ leaf_start,
                                                        const int* __restrict__ leaf_count, int n_leaves, uint2* __restrict__ cell_range,
                                                        int div_x, int* __restrict__ row_any) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_leaves; i += gridDim.x * kBlock) {
    const int c = leaf_cell[i];
    // the cell's segment of the cell-ordered points, in the table itself: a probe that finds an occupied cell needs
    // no second dependent lookup (on a 40 MB table of a 1M-point target that lookup is another trip to HBM)
    cell_range[c] = make_uint2(leaf_start[i], static_cast<unsigned>(leaf_count[i]));
    row_any[c / div_x] = 1;  // the x-row (y, z) of this cell holds points: empty rows are skipped by the shell search
  }
}

// Points in cell order (sorted_pts[q] = pts[sorted_idx[q]]): the candidates of a cell are consecutive
// 16-byte records, so a scan issues several loads at once instead of chasing index -> point one
// candidate at a time (with ~16k queries the chip is nearly empty and a query's time is its chain of
// load latencies).
__global__ __launch_bounds__(kBlock) void k_gather_points(const float4* __restrict__ pts, const int* __restrict__ sorted_idx,
                                                          const unsigned* __restrict__ d_n_sorted, float4* __restrict__ out) {
  const int n = static_cast<int>(*d_n_sorted);
  for (int q = blockIdx.x * kBlock + threadIdx.x; q < n; q += gridDim.x * kBlock) out[q] = pts[sorted_idx[q]];
}


__global__ __launch_bounds__(kBlock) void k_fitness(const float4* __restrict__ src, int n, EvalParams P, PointIndex ix,
                                                    double max_range, double* __restrict__ partials) {
  constexpr int kTeams = kBlock / kTeam;
  __shared__ double lds[(kBlock / kWave) * 32];
  double acc[kNumAcc];
#pragma unroll
  for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
  const int sub = threadIdx.x & (kTeam - 1);
  const float leaf = fminf(ix.geom.leaf[0], fminf(ix.geom.leaf[1], ix.geom.leaf[2]));
  const int r_lim = max(ix.geom.div_b[0], max(ix.geom.div_b[1], ix.geom.div_b[2]));
  const int r_max = max_shells(ix, r_lim);
  // The loop is uniform across the WAVE (a team without a query idles): the fallback below is a wave-wide operation.
  constexpr int kTeamsPerWave = kWave / kTeam;
  const int wave_in_block = threadIdx.x / kWave, team_in_wave = (threadIdx.x & (kWave - 1)) / kTeam;
  // Queries are dealt to the teams in BIT-REVERSED order: the eight teams of a wave take queries an eighth of the scan apart.
  // The far queries of a scan come in runs (a wall the target does not cover), and a wave that held eight of them was the
  // kernel's whole duration while the rest of the chip sat idle.
  const int log_slots = 32 - __clz(max(n, 2) - 1), n_slots = 1 << log_slots;
  for (int base = (blockIdx.x * (kBlock / kWave) + wave_in_block) * kTeamsPerWave; base < n_slots; base += gridDim.x * kTeams) {
    const int i = static_cast<int>(__brev(static_cast<unsigned>(base + team_in_wave)) >> (32 - log_slots));
    float tx = 0.f, ty = 0.f, tz = 0.f;
    bool live = i < n;  // uniform within a team
    if (live) {
      const float4 pt = src[i];
      live = finite3(pt.x, pt.y, pt.z);  // transformPointCloud leaves it non-finite; no neighbour to report
      if (live) {
        xform_point(P.T, pt.x, pt.y, pt.z, tx, ty, tz);
        live = finite3(tx, ty, tz);
      }
    }
    float tb = INFINITY;
    bool done = !live;
    int ci = 0, cj = 0, ck = 0;
    float margin = 0.0f;
    // shells 0 .. kTeamShells by the query's team of 8 lanes (a registered scan: nearly every query ends in its own cell or
    // the 26 around it) ...
    constexpr int kTeamShells = 2;
    if (live) {
      float best = INFINITY;  // this lane's share of the candidates
      auto consider = [&](float d, unsigned, bool ok) {
        if (ok) best = fminf(best, d);
      };
      query_cell(ix.geom, tx, ty, tz, ci, cj, ck, margin);
      for (int r = 0; r <= min(r_max, kTeamShells) && !done; r++) {
        team_shell(ix, ci, cj, ck, r, sub, tx, ty, tz, consider, tb);  // (tb: the team's best after the previous shell)
        tb = best;
#pragma unroll
        for (int off = 1; off < kTeam; off <<= 1) tb = fminf(tb, __shfl_xor(tb, off, kWave));
        // every unvisited cell is at least r cells away (less the slack for the build-time / search-time
        // index rounding, trap 2)
        const float reach = static_cast<float>(r) * leaf + margin - ix.slack;
        if ((reach > 0.0f && tb <= reach * reach) || r >= r_lim) done = true;
      }
    }
    // ... the far queries (parts of the scan the target does not cover: metres to the nearest point, shells of hundreds of
    // rows) by the WHOLE wave, one unfinished query at a time: 64 rows of a shell per step instead of 8.  602 such queries
    // of the reference pair's 15 950 were 2/3 of this kernel's time when their teams walked the shells alone.
    unsigned long long open_teams = __ballot(!done && sub == 0);
    const int lane = threadIdx.x & (kWave - 1);
    while (open_teams) {
      const int src_lane = __builtin_ctzll(open_teams);
      open_teams &= open_teams - 1;
      const float qx = __shfl(tx, src_lane, kWave), qy = __shfl(ty, src_lane, kWave), qz = __shfl(tz, src_lane, kWave);
      const int qi = __shfl(ci, src_lane, kWave), qj = __shfl(cj, src_lane, kWave), qk = __shfl(ck, src_lane, kWave);
      const float q_margin = __shfl(margin, src_lane, kWave);
      float wb = __shfl(tb, src_lane, kWave);  // what the team has found so far bounds the search
      float wbest = INFINITY;
      auto consider_w = [&](float d, unsigned, bool ok) {
        if (ok) wbest = fminf(wbest, d);
      };
      bool wdone = false;
      for (int r = kTeamShells + 1; r <= r_max && !wdone; r++) {
        team_shell<kWave>(ix, qi, qj, qk, r, lane, qx, qy, qz, consider_w, wb);
        float m = wbest;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) m = fminf(m, __shfl_xor(m, off, kWave));
        wb = fminf(wb, m);
        const float reach = static_cast<float>(r) * leaf + q_margin - ix.slack;
        if ((reach > 0.0f && wb <= reach * reach) || r >= r_lim) wdone = true;
      }
      if (!wdone) {  // nothing within r_max shells: one scan over all points
        float wd;
        int wi;
        wave_nearest(ix, qx, qy, qz, wd, wi);
        wb = wd;
      }
      if (lane / kTeam == src_lane / kTeam) tb = wb;
    }
    if (live && sub == 0 && static_cast<double>(tb) <= max_range) {  // the squared distance against max_range, as PCL does
      acc[0] += static_cast<double>(tb);
      acc[1] += 1.0;
    }
  }
  block_reduce_store<kNumAcc>(acc, partials + static_cast<size_t>(blockIdx.x) * kEvalStride, lds);
}

// calculateScore (ndt_omp_impl.hpp:935-983): cloud used as given, f64 throughout
template <int NNB>
__global__ __launch_bounds__(kBlock) void k_calc_score(const float4* __restrict__ cloud, int n, GridView gv, double d1,
                                                       double d2, double d3, float r2, double* __restrict__ partials) {
  __shared__ double lds[(kBlock / kWave) * 32];
  double acc[1] = {0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 pt = cloud[i];
    if (!finite3(pt.x, pt.y, pt.z)) continue;  // no neighbourhood in the reference (garbage voxel index): adds nothing
    int vi, vj, vk;
    search_ijk(gv.g, pt.x, pt.y, pt.z, vi, vj, vk);
    if (!near_grid(gv.g, vi, vj, vk)) continue;
    const unsigned centre = lut_index(gv.g, vi, vj, vk);
    int rec[NNB];
    int cnt = 0;
    for (int k = 0; k < NNB; k++) {
      int dx, dy, dz;
      nb_offset<NNB>(k, dx, dy, dz);
      rec[k] = (NNB == 27) ? probe_kd(gv, vi, vj, vk, centre, dx, dy, dz, pt.x, pt.y, pt.z, r2) : probe(gv, vi, vj, vk, centre, dx, dy, dz);
      cnt += (rec[k] >= 0);
    }
    for (int k = 0; k < NNB; k++) {
      if (rec[k] < 0) continue;
      const double* mu = gv.recs[rec[k]].mean;
      const double* ic = gv.centroids[rec[k]].icov;  // the leaf's f64 icov_ (:966)
      const double x0 = static_cast<double>(pt.x) - mu[0], x1 = static_cast<double>(pt.y) - mu[1], x2 = static_cast<double>(pt.z) - mu[2];
      const double c00 = ic[0], c01 = ic[1], c02 = ic[2], c11 = ic[3], c12 = ic[4], c22 = ic[5];
      const double c0 = (c00 * x0 + c01 * x1) + c02 * x2;
      const double c1 = (c01 * x0 + c11 * x1) + c12 * x2;
      const double c2 = (c02 * x0 + c12 * x1) + c22 * x2;
      const double e = exp(-d2 * ((x0 * c0 + x1 * c1) + x2 * c2) / 2);
      acc[0] += (-d1 * e - d3) / cnt;
    }
  }
  block_reduce_store<1>(acc, partials + static_cast<size_t>(blockIdx.x) * kEvalStride, lds);
}

inline int grid_for(size_t n, int max_blocks) {
  size_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > static_cast<size_t>(max_blocks)) b = max_blocks;
  return static_cast<int>(b);
}

}  // namespace

// ===========================================================================
// launchers
// ===========================================================================
// Tunable (development aid): NDT_K2_MAX_BLOCKS caps the grid.
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
int derivative_variant() { return 0; }  // (the development variants of round 1 are gone: one body, spelled out by hand)
// blocks a scan of n points is walked by inside a lock-step batch: two points per thread (the wave fold and the block
// epilogue are paid once per thread), a function of the scan's size only
int batch_blocks(int n) { return max(1, (n + kBatchPointsPerBlock - 1) / kBatchPointsPerBlock); }
int derivative_blocks(int n, int search) {
  static const int cap = env_int("NDT_K2_MAX_BLOCKS", 1024);
  (void)search;
  return grid_for(static_cast<size_t>(n), cap);
}

hipError_t launch_selftest_reduce(int n_blocks, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(k_selftest_reduce, dim3(n_blocks), dim3(kBlock), 0, stream, out);
  return hipGetLastError();
}

hipError_t launch_derivatives_stamped(const float4* src, int n, const GridView& gv, const EvalParams& P, int n_blocks,
                                      double* partials, unsigned long long* stamps, hipStream_t stream) {
  hipLaunchKernelGGL(k_derivatives_stamped, dim3(n_blocks), dim3(kBlock), 0, stream, src, n, gv, P, partials, stamps);
  return hipGetLastError();
}

template <int NNB, bool WANT_H>
static void launch_deriv_t(const float4* src, int n, const GridView& gv, const EvalParams& P, const ScanDesc* descs,
                           const int* active, int n_active, int max_blocks, int n_blocks, double* partials,
                           hipStream_t stream) {
  if (descs)
    hipLaunchKernelGGL((k_derivatives<NNB, WANT_H, true>), dim3(n_blocks, n_active), dim3(kBlock), 0, stream, src,
                       n, gv, P, descs, active, max_blocks, partials);
  else
    hipLaunchKernelGGL((k_derivatives<NNB, WANT_H, false>), dim3(n_blocks, 1), dim3(kBlock), 0, stream, src, n,
                       gv, P, descs, active, max_blocks, partials);
}

hipError_t launch_derivatives(const float4* src, int n, const GridView& gv, const EvalParams& P, int search,
                              bool want_hessian, const ScanDesc* descs, const int* active, int n_active, int max_blocks,
                              int n_blocks, double* partials, hipStream_t stream) {
  // search: 0 = KDTREE, 1 = DIRECT26, 2 = DIRECT7 (and the reference's `default:`), 3 = DIRECT1
#define NDT_LAUNCH_DERIV(NNB, H) launch_deriv_t<NNB, H>(src, n, gv, P, descs, active, n_active, max_blocks, n_blocks, partials, stream)
  if (search == 0) {
    if (want_hessian) NDT_LAUNCH_DERIV(27, true); else NDT_LAUNCH_DERIV(27, false);
  } else if (search == 1) {
    if (want_hessian) NDT_LAUNCH_DERIV(26, true); else NDT_LAUNCH_DERIV(26, false);
  } else if (search == 3) {
    if (want_hessian) NDT_LAUNCH_DERIV(1, true); else NDT_LAUNCH_DERIV(1, false);
  } else {
    if (want_hessian) NDT_LAUNCH_DERIV(7, true); else NDT_LAUNCH_DERIV(7, false);
  }
#undef NDT_LAUNCH_DERIV
  return hipGetLastError();
}

hipError_t launch_batch_step(const float4* src, const GridView& gv, int search, const ScanDesc* descs, const int* active,
                             int n_active, int max_blocks, int n_blocks, double* partials, hipStream_t stream) {
  const dim3 grid(n_blocks, n_active), block(kBlock);
  if (search == 0) hipLaunchKernelGGL(k_batch_step<27>, grid, block, 0, stream, src, gv, descs, active, max_blocks, partials);
  else if (search == 1) hipLaunchKernelGGL(k_batch_step<26>, grid, block, 0, stream, src, gv, descs, active, max_blocks, partials);
  else if (search == 3) hipLaunchKernelGGL(k_batch_step<1>, grid, block, 0, stream, src, gv, descs, active, max_blocks, partials);
  else hipLaunchKernelGGL(k_batch_step<7>, grid, block, 0, stream, src, gv, descs, active, max_blocks, partials);
  return hipGetLastError();
}

template <int NNB>
static void launch_h64_t(const float4* src, int n, const GridView& gv, const Hess64Params& P, const ScanDesc* descs,
                         const int* active, int n_active, int max_blocks, int n_blocks, double* partials,
                         hipStream_t stream) {
  if (descs)
    hipLaunchKernelGGL((k_hessian64<NNB, true>), dim3(n_blocks, n_active), dim3(kBlock), 0, stream, src, n, gv, P, descs,
                       active, max_blocks, partials);
  else
    hipLaunchKernelGGL((k_hessian64<NNB, false>), dim3(n_blocks, 1), dim3(kBlock), 0, stream, src, n, gv, P, descs, active,
                       max_blocks, partials);
}

hipError_t launch_hessian64(const float4* src, int n, const GridView& gv, const Hess64Params& P, int search,
                            const ScanDesc* descs, const int* active, int n_active, int max_blocks, int n_blocks,
                            double* partials, hipStream_t stream) {
  if (search == 0) launch_h64_t<27>(src, n, gv, P, descs, active, n_active, max_blocks, n_blocks, partials, stream);
  else if (search == 1) launch_h64_t<26>(src, n, gv, P, descs, active, n_active, max_blocks, n_blocks, partials, stream);
  else if (search == 3) launch_h64_t<1>(src, n, gv, P, descs, active, n_active, max_blocks, n_blocks, partials, stream);
  else launch_h64_t<7>(src, n, gv, P, descs, active, n_active, max_blocks, n_blocks, partials, stream);
  return hipGetLastError();
}

hipError_t launch_reduce(const double* partials, int n_blocks, int n_scans, const ScanDesc* descs, double* out,
                         hipStream_t stream, unsigned long long seq) {
  hipLaunchKernelGGL(k_reduce, dim3(n_scans), dim3(kReduceThreads), 0, stream, partials, n_blocks, descs, out, seq);
  return hipGetLastError();
}

hipError_t launch_publish_rows(const double* d_rows, int n_rows, double* host_rows, unsigned long long seq, hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_publish_rows, dim3(n_rows), dim3(kWave), 0, stream, d_rows, host_rows, seq);
  return hipGetLastError();
}

hipError_t launch_transform(const float4* src, int n, const float* T12, float4* dst, hipStream_t stream, int dense) {
  if (n == 0) return hipSuccess;
  EvalParams P = {};
  for (int i = 0; i < 12; i++) P.T[i] = T12[i];
  hipLaunchKernelGGL(k_transform, dim3(grid_for(n, 2048)), dim3(kBlock), 0, stream, src, n, P, dst, dense);
  return hipGetLastError();
}

hipError_t launch_calc_score(const float4* cloud, int n, const GridView& gv, double d1, double d2, double d3, int search,
                             float r2, int n_blocks, double* partials, hipStream_t stream) {
  if (search == 0)
    hipLaunchKernelGGL(k_calc_score<27>, dim3(n_blocks), dim3(kBlock), 0, stream, cloud, n, gv, d1, d2, d3, r2, partials);
  else if (search == 1)
    hipLaunchKernelGGL(k_calc_score<26>, dim3(n_blocks), dim3(kBlock), 0, stream, cloud, n, gv, d1, d2, d3, r2, partials);
  else if (search == 3)
    hipLaunchKernelGGL(k_calc_score<1>, dim3(n_blocks), dim3(kBlock), 0, stream, cloud, n, gv, d1, d2, d3, r2, partials);
  else
    hipLaunchKernelGGL(k_calc_score<7>, dim3(n_blocks), dim3(kBlock), 0, stream, cloud, n, gv, d1, d2, d3, r2, partials);
  return hipGetLastError();
}

hipError_t launch_cell_ranges(const int* leaf_cell, const unsigned* leaf_start, const int* leaf_count, int n_leaves, uint2* cell_range,
                              int div_x, int* row_any, hipStream_t stream) {
  hipLaunchKernelGGL(k_cell_ranges, dim3(grid_for(n_leaves, 1024)), dim3(kBlock), 0, stream, leaf_cell, leaf_start, leaf_count, n_leaves,
                     cell_range, div_x, row_any);
  return hipGetLastError();
}

hipError_t launch_fitness(const float4* src, int n, const float* T12, const PointIndex& tgt, double max_range, int n_blocks,
                          double* partials, hipStream_t stream) {
  EvalParams P{};
  for (int i = 0; i < 12; i++) P.T[i] = T12[i];
  hipLaunchKernelGGL(k_fitness, dim3(n_blocks), dim3(kBlock), 0, stream, src, n, P, tgt, max_range, partials);
  return hipGetLastError();
}

hipError_t launch_gather_points(const float4* pts, const int* sorted_idx, const unsigned* d_n_sorted, int n_max, float4* out,
                                hipStream_t stream) {
  const int blocks = max(1, min(2048, (n_max + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(k_gather_points, dim3(blocks), dim3(kBlock), 0, stream, pts, sorted_idx, d_n_sorted, out);
  return hipGetLastError();
}

}  // namespace ndt
