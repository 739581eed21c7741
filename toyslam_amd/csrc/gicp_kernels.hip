// gicp_kernels.hip -- HIP kernels of the GICP row, gfx950 / wave64 only.
//
//   k_knn_covariances  computeCovariances (gicp_omp_impl.hpp:48-116): exact k nearest neighbours of
//                      every point of a cloud among the cloud itself, the f64 covariance of those k
//                      points, its eigenvectors, and the regularised covariance (1, 1, epsilon).
//   k_correspond       per outer iteration (:419-456): nearest target point of every transformed
//                      source point, distance gate, Mahalanobis matrix (R C1 R^T + C2)^-1.
//   k_functor          the BFGS objective / gradient sums (:241-368), one fused launch per evaluation.
//
// All three are gather kernels over point sets that sit in L2 at the reference's sizes (tens of
// thousands of points after the 0.1 m prefilter, apps/align.cpp:60-69); nothing here is
// GEMM-shaped.  Nearest-neighbour search: the target's counting-sort voxel index (K1), cubic
// shells of cells around the query until the k-th best distance cannot be beaten by any unvisited
// shell -- exact, and with (distance, index) ordering deterministic.
#include "gicp_kernels.hpp"

#include "ndt_device.hpp"
#include "ndt_search.hpp"

namespace gicp {

using namespace ndt;

namespace {

constexpr int kKnnBlock = 64;   // one wave = 8 query teams per block

// [Eigen] Matrix4f * Vector4f (column by column): row r = ((T_r0 x + T_r1 y) + T_r2 z) + T_r3 * 1
__device__ __forceinline__ void matvec_eigen(const float* T12, float x, float y, float z, float& ox, float& oy, float& oz) {
#pragma clang fp contract(off)
  ox = ((T12[0] * x + T12[1] * y) + T12[2] * z) + T12[3] * 1.0f;
  oy = ((T12[4] * x + T12[5] * y) + T12[6] * z) + T12[7] * 1.0f;
  oz = ((T12[8] * x + T12[9] * y) + T12[10] * z) + T12[11] * 1.0f;
}

__global__ __launch_bounds__(kKnnBlock) void k_knn_covariances(PointIndex ix, int k, double gicp_epsilon,
                                                               double* __restrict__ cov6, int* __restrict__ nn_idx,
                                                               float* __restrict__ nn_d2) {
#pragma clang fp contract(off)
  // One candidate list PER TEAM (k entries, unordered, its worst entry tracked), not one per lane: an eighth of the
  // LDS (the per-lane lists capped the kernel at 3.5 waves per SIMD) and a bound every lane prunes with.  A candidate
  // that beats the worst entry replaces it and the team finds the new worst together; order is established once, at the
  // end, by a rank sort.  Entries are POSITIONS in the cell order; the point index behind one is looked up only when
  // two distances are equal (the tie rule is on the index) and at the end.
  constexpr int kTeams = kKnnBlock / kTeam;
  extern __shared__ unsigned char knn_lds[];
  const int lane = threadIdx.x, sub = lane & (kTeam - 1), team = lane / kTeam, team_base = lane & ~(kTeam - 1);
  float* ld = reinterpret_cast<float*>(knn_lds) + team * k;                    // [8][k] distances
  int* lp = reinterpret_cast<int*>(knn_lds) + kTeams * k + team * k;           // [8][k] positions
  int* ordered = reinterpret_cast<int*>(knn_lds) + 2 * kTeams * k + team * k;  // [8][k] point indices, ascending (distance, index)
  const float leaf = fminf(ix.geom.leaf[0], fminf(ix.geom.leaf[1], ix.geom.leaf[2]));
  const int r_lim = max(ix.geom.div_b[0], max(ix.geom.div_b[1], ix.geom.div_b[2]));
  const int r_max = max_shells(ix, r_lim);
  auto lds_fence = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };  // one wave per block: program order is enough
  // The loop is uniform across the wave (a team without a query idles): the scan of the isolated queries is wave-wide.
  for (int base = blockIdx.x * kTeams; base < ix.n; base += gridDim.x * kTeams) {
    const int i = base + team;  // uniform within a team
    const bool live = i < ix.n;
    const float4 q = live ? ix.pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;  // entries in the list (team-uniform, like everything below that is not marked "lane")
    float worst = INFINITY;
    int worst_p = 0, worst_slot = 0;
    auto before = [&](float da, int pa, float db, int pb) {  // (distance, index) order
      return da < db || (da == db && ix.sorted_idx[pa] < ix.sorted_idx[pb]);
    };
    auto find_worst = [&] {  // the list's last entry in (distance, index) order
      float bd = -1.0f;  // lane: best so far over my slots
      int bp = 0, bs = -1;
      for (int j = sub; j < cnt; j += kTeam) {
        const float d = ld[j];
        const int p = lp[j];
        if (bs < 0 || before(bd, bp, d, p)) {
          bd = d;
          bp = p;
          bs = j;
        }
      }
#pragma unroll
      for (int off = 1; off < kTeam; off <<= 1) {
        const float od = __shfl_xor(bd, off, kWave);
        const int op = __shfl_xor(bp, off, kWave), os = __shfl_xor(bs, off, kWave);
        if (os >= 0 && (bs < 0 || before(bd, bp, od, op))) {
          bd = od;
          bp = op;
          bs = os;
        }
      }
      worst = bd;
      worst_p = bp;
      worst_slot = bs;
    };
    auto insert_one = [&](float cd, int cp) {  // team-collective: a candidate known to all eight lanes
      if (cnt < k) {
        if (sub == 0) {
          ld[cnt] = cd;
          lp[cnt] = cp;
        }
        lds_fence();
        cnt++;
        if (cnt == k) find_worst();
      } else if (before(cd, cp, worst, worst_p)) {
        if (sub == 0) {
          ld[worst_slot] = cd;
          lp[worst_slot] = cp;
        }
        lds_fence();
        find_worst();
      }
    };
    // every lane brings one candidate (or none); the ones that can enter the list are taken one at a time
    auto offer = [&](float d, unsigned upos, bool ok) {
      const int pos = static_cast<int>(upos);
      const bool want = ok && (cnt < k || before(d, pos, worst, worst_p));
      unsigned pending = static_cast<unsigned>((__ballot(want) >> team_base) & 0xffull);
      while (pending) {
        const int owner = __builtin_ctz(pending);
        pending &= pending - 1;
        insert_one(__shfl(d, team_base + owner, kWave), __shfl(pos, team_base + owner, kWave));
      }
    };
    int ci, cj, ck;
    float margin;
    query_cell(ix.geom, q.x, q.y, q.z, ci, cj, ck, margin);
    bool done = !live;
    for (int r = 0; r <= r_max && !done; r++) {
      team_shell(ix, ci, cj, ck, r, sub, q.x, q.y, q.z, offer);
      // Every unvisited point is at least r cells plus the query's margin (less the index-rounding slack) away: once the
      // k-th best lies strictly inside that reach no unvisited point can enter the k nearest.
      const float reach = static_cast<float>(r) * leaf + margin - ix.slack;
      if ((cnt == k && reach > 0.0f && worst < reach * reach) || r >= r_lim) done = true;
    }
    // Sparse neighbourhoods: the WAVE looks at every point, twice, for one unfinished query at a time.  Pass 1 only keeps
    // each lane's eight smallest distances, in registers; the k-th smallest of the wave's 512 values bounds the k-th
    // neighbour's distance from above.  Pass 2 hands the points inside that bound -- about k of them -- to the query's team,
    // which fills its list afresh.  (The few isolated queries set this kernel's time when a team walks the cloud alone,
    // and a single pass with the list spends its time updating it: points arrive in arbitrary order.)
    unsigned long long open_teams = __ballot(!done && sub == 0);
    while (open_teams) {
      const int src_lane = __builtin_ctzll(open_teams);
      open_teams &= open_teams - 1;
      const bool mine = (team == src_lane / kTeam);
      const float ox = __shfl(q.x, src_lane, kWave), oy = __shfl(q.y, src_lane, kWave), oz = __shfl(q.z, src_lane, kWave);
      const int count = ix.n_sorted;
      float m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY, m4 = INFINITY, m5 = INFINITY, m6 = INFINITY, m7 = INFINITY;
      auto keep = [&](float d, bool ok) {
        if (!ok || !(d < m7)) return;
        m7 = d;  // bubble the newcomer down the sorted registers
        float t;
        if (m7 < m6) { t = m6; m6 = m7; m7 = t; }
        if (m6 < m5) { t = m5; m5 = m6; m6 = t; }
        if (m5 < m4) { t = m4; m4 = m5; m5 = t; }
        if (m4 < m3) { t = m3; m3 = m4; m4 = t; }
        if (m3 < m2) { t = m2; m2 = m3; m3 = t; }
        if (m2 < m1) { t = m1; m1 = m2; m2 = t; }
        if (m1 < m0) { t = m0; m0 = m1; m1 = t; }
      };
      for (int b0 = 0; b0 < count; b0 += 4 * kWave) {
        const int p0 = b0 + lane, p1 = p0 + kWave, p2 = p1 + kWave, p3 = p2 + kWave;
        const float4 a = ix.sorted_pts[min(p0, count - 1)], b = ix.sorted_pts[min(p1, count - 1)];
        const float4 c = ix.sorted_pts[min(p2, count - 1)], d = ix.sorted_pts[min(p3, count - 1)];
        keep(dist2_f32(ox, oy, oz, a.x, a.y, a.z), p0 < count);
        keep(dist2_f32(ox, oy, oz, b.x, b.y, b.z), p1 < count);
        keep(dist2_f32(ox, oy, oz, c.x, c.y, c.z), p2 < count);
        keep(dist2_f32(ox, oy, oz, d.x, d.y, d.z), p3 < count);
      }
      float bound = INFINITY;
      for (int j = 0; j < k; j++) {  // pop the wave's smallest head k times (n >= k values exist)
        float h = m0;
        int who = lane;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
          const float oh = __shfl_xor(h, off, kWave);
          const int ow = __shfl_xor(who, off, kWave);
          if (oh < h || (oh == h && ow < who)) {
            h = oh;
            who = ow;
          }
        }
        bound = h;
        if (who == lane) { m0 = m1; m1 = m2; m2 = m3; m3 = m4; m4 = m5; m5 = m6; m6 = m7; m7 = INFINITY; }
      }
      if (mine) {
        cnt = 0;
        worst = INFINITY;
      }
      auto hand_over = [&](float d, int pos, bool ok) {  // wave-uniform call: the hits go to the owning team one by one
        unsigned long long hits = __ballot(ok && !(d > bound));
        while (hits) {
          const int from = __builtin_ctzll(hits);
          hits &= hits - 1;
          const float cd = __shfl(d, from, kWave);
          const int cp = __shfl(pos, from, kWave);
          if (mine) insert_one(cd, cp);
        }
      };
      for (int b0 = 0; b0 < count; b0 += 4 * kWave) {
        const int p0 = b0 + lane, p1 = p0 + kWave, p2 = p1 + kWave, p3 = p2 + kWave;
        const float4 a = ix.sorted_pts[min(p0, count - 1)], b = ix.sorted_pts[min(p1, count - 1)];
        const float4 c = ix.sorted_pts[min(p2, count - 1)], d = ix.sorted_pts[min(p3, count - 1)];
        hand_over(dist2_f32(ox, oy, oz, a.x, a.y, a.z), p0, p0 < count);
        hand_over(dist2_f32(ox, oy, oz, b.x, b.y, b.z), p1, p1 < count);
        hand_over(dist2_f32(ox, oy, oz, c.x, c.y, c.z), p2, p2 < count);
        hand_over(dist2_f32(ox, oy, oz, d.x, d.y, d.z), p3, p3 < count);
      }
    }
    if (!live) continue;
    // rank sort: entry j goes to the place given by the number of entries before it
    for (int j = sub; j < cnt; j += kTeam) {
      const float d = ld[j];
      const int p = lp[j];
      int rank = 0;
      for (int t = 0; t < cnt; t++) rank += (t != j && before(ld[t], lp[t], d, p)) ? 1 : 0;
      const int idx = ix.sorted_idx[p];
      ordered[rank] = idx;
      if (nn_idx) {
        nn_idx[static_cast<size_t>(i) * k + rank] = idx;
        nn_d2[static_cast<size_t>(i) * k + rank] = d;
      }
    }
    if (nn_idx)
      for (int j = cnt + sub; j < k; j += kTeam) {  // (only a cloud smaller than k, which the host refuses)
        nn_idx[static_cast<size_t>(i) * k + j] = -1;
        nn_d2[static_cast<size_t>(i) * k + j] = INFINITY;
      }
    lds_fence();
    if (sub != 0) continue;  // the 3x3 algebra of a query is one lane's work
    // :81-105  f32 products, f64 sums, neighbours in ascending distance
    double mx = 0, my = 0, mz = 0, cxx = 0, cyx = 0, cyy = 0, czx = 0, czy = 0, czz = 0;
    auto add = [&](const float4& t) {
      mx += static_cast<double>(t.x);
      my += static_cast<double>(t.y);
      mz += static_cast<double>(t.z);
      cxx += static_cast<double>(t.x * t.x);
      cyx += static_cast<double>(t.y * t.x);
      cyy += static_cast<double>(t.y * t.y);
      czx += static_cast<double>(t.z * t.x);
      czy += static_cast<double>(t.z * t.y);
      czz += static_cast<double>(t.z * t.z);
    };
    auto fetch = [&](int j) { return ix.pts[min(max(ordered[j], 0), ix.n - 1)]; };
    int j = 0;
    for (; j + 4 <= cnt; j += 4) {  // four gathers in flight, added in order
      const float4 t0 = fetch(j), t1 = fetch(j + 1), t2 = fetch(j + 2), t3 = fetch(j + 3);
      add(t0);
      add(t1);
      add(t2);
      add(t3);
    }
    for (; j < cnt; j++) add(fetch(j));
    const double kd = static_cast<double>(k);
    mx /= kd;
    my /= kd;
    mz /= kd;
    double C[3][3];
    C[0][0] = cxx / kd - mx * mx;
    C[1][0] = cyx / kd - my * mx;
    C[1][1] = cyy / kd - my * my;
    C[2][0] = czx / kd - mz * mx;
    C[2][1] = czy / kd - mz * my;
    C[2][2] = czz / kd - mz * mz;
    C[0][1] = C[1][0];
    C[0][2] = C[2][0];
    C[1][2] = C[2][1];
    // :108-120  [Eigen] JacobiSVD of a symmetric matrix: U = eigenvectors ordered by |eigenvalue| descending
    double w[3], V[3][3];
    eig3_jacobi(C, w, V);
    int o0 = 2, o1 = 1, o2 = 0;  // eig3_jacobi: ascending eigenvalues
    if (fabs(w[o1]) > fabs(w[o0])) { const int t = o0; o0 = o1; o1 = t; }
    if (fabs(w[o2]) > fabs(w[o1])) { const int t = o1; o1 = o2; o2 = t; }
    if (fabs(w[o1]) > fabs(w[o0])) { const int t = o0; o0 = o1; o1 = t; }
    double out[6];
    int e = 0;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) {
        double s = (1.0 * V[a][o0]) * V[b][o0];
        s += (1.0 * V[a][o1]) * V[b][o1];
        s += (gicp_epsilon * V[a][o2]) * V[b][o2];
        out[e++] = s;
      }
    for (int t = 0; t < 6; t++) cov6[static_cast<size_t>(i) * 6 + t] = out[t];
  }
}

__device__ __forceinline__ void load_sym(const double* __restrict__ c6, double C[3][3]) {
  C[0][0] = c6[0]; C[0][1] = c6[1]; C[0][2] = c6[2];
  C[1][0] = c6[1]; C[1][1] = c6[3]; C[1][2] = c6[4];
  C[2][0] = c6[2]; C[2][1] = c6[4]; C[2][2] = c6[5];
}

__global__ __launch_bounds__(kBlock) void k_correspond(const float4* __restrict__ output, int n, EvalParams P, Rot3d R,
                                                       PointIndex ix, const double* __restrict__ cov_src6,
                                                       const double* __restrict__ cov_tgt6, double dist_threshold,
                                                       int* __restrict__ corr, float* __restrict__ maha9) {
#pragma clang fp contract(off)
  constexpr int kTeams = kBlock / kTeam;
  const int sub = threadIdx.x & (kTeam - 1);
  const float leaf = fminf(ix.geom.leaf[0], fminf(ix.geom.leaf[1], ix.geom.leaf[2]));
  const int r_lim = max(ix.geom.div_b[0], max(ix.geom.div_b[1], ix.geom.div_b[2]));
  const int r_max = max_shells(ix, r_lim);
  // The loop is uniform across the WAVE (a team without a query idles): the fallback below is a wave-wide operation.
  constexpr int kTeamsPerWave = kWave / kTeam;
  const int wave_in_block = threadIdx.x / kWave, team_in_wave = (threadIdx.x & (kWave - 1)) / kTeam;
  for (int base = (blockIdx.x * (kBlock / kWave) + wave_in_block) * kTeamsPerWave; base < n; base += gridDim.x * kTeams) {
    const int i = base + team_in_wave;
    const bool live = i < n;  // uniform within a team
    float qx = 0.f, qy = 0.f, qz = 0.f;
    bool done = !live;
    float tb = INFINITY;  // the team's best
    int tb_i = 0x7fffffff;
    if (live) {
      const float4 p = output[i];
      matvec_eigen(P.T, p.x, p.y, p.z, qx, qy, qz);
      float best = INFINITY;  // this lane's share: distance and POSITION in the cell order (the index behind it is
      int best_p = -1;        // looked up on ties and at the end of a shell only)
      auto consider = [&](float d, unsigned pos, bool ok) {
        if (!ok || d > best) return;
        if (d < best || ix.sorted_idx[pos] < ix.sorted_idx[best_p]) {  // equal distance: the lower index
          best = d;
          best_p = static_cast<int>(pos);
        }
      };
      int ci, cj, ck;
      float margin;
      query_cell(ix.geom, qx, qy, qz, ci, cj, ck, margin);
      for (int r = 0; r <= r_max && !done; r++) {
        team_shell(ix, ci, cj, ck, r, sub, qx, qy, qz, consider);
        tb = best;
        tb_i = best_p >= 0 ? ix.sorted_idx[best_p] : 0x7fffffff;
        team_min(tb, tb_i);
        const float reach = static_cast<float>(r) * leaf + margin - ix.slack;
        if ((tb_i != 0x7fffffff && reach > 0.0f && tb < reach * reach) || r >= r_lim) done = true;
        // nothing closer than the gate is left once the shells reach past it: no correspondence either way
        if (reach > 0.0f && static_cast<double>(reach) * static_cast<double>(reach) >= dist_threshold &&
            !(static_cast<double>(tb) < dist_threshold))
          done = true;
      }
    }
    // sparse neighbourhoods: the wave scans everything, one unfinished query at a time
    unsigned long long open_teams = __ballot(!done && sub == 0);
    while (open_teams) {
      const int src_lane = __builtin_ctzll(open_teams);
      open_teams &= open_teams - 1;
      float wd;
      int wi;
      wave_nearest(ix, __shfl(qx, src_lane, kWave), __shfl(qy, src_lane, kWave), __shfl(qz, src_lane, kWave), wd, wi);
      if (team_in_wave == src_lane / kTeam) {
        tb = wd;
        tb_i = wi;
      }
    }
    if (!live) continue;
    if (sub != 0) continue;
    int c_out = -1;
    if (tb_i != 0x7fffffff && static_cast<double>(tb) < dist_threshold) {  // :436
      double C1[3][3], C2[3][3], M[3][3], tmp[3][3], inv[3][3];
      load_sym(cov_src6 + static_cast<size_t>(i) * 6, C1);
      load_sym(cov_tgt6 + static_cast<size_t>(tb_i) * 6, C2);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) M[r][c] = (R.m[r * 3] * C1[0][c] + R.m[r * 3 + 1] * C1[1][c]) + R.m[r * 3 + 2] * C1[2][c];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
          tmp[r][c] = ((M[r][0] * R.m[c * 3] + M[r][1] * R.m[c * 3 + 1]) + M[r][2] * R.m[c * 3 + 2]) + C2[r][c];
      inv3_cofactor(tmp, inv);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) maha9[static_cast<size_t>(i) * 9 + r * 3 + c] = static_cast<float>(inv[r][c]);
      c_out = tb_i;
    }
    corr[i] = c_out;
  }
}

// MODE 0: operator() only; 1: df / fdf (f64); 2: operator() in slot 0 AND the f64 gradient sums of df in slots 1..12 --
// the line search asks for df right after operator() at the same point, and one launch serves both
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_functor(const float4* __restrict__ output, int n, const float4* __restrict__ tgt,
                                                    const int* __restrict__ corr, const float* __restrict__ maha9,
                                                    EvalParams P, double* __restrict__ partials,
                                                    unsigned* __restrict__ counter, double* __restrict__ out_row,
                                                    unsigned long long seq) {
#pragma clang fp contract(off)
  constexpr int kWaves = kBlock / kWave, kParts = kBlock / kEvalStride;
  __shared__ double lds[kWaves * 32];
  __shared__ double lds2[kParts * kEvalStride];
  __shared__ int s_last;
  double acc[kFunctorValues];
#pragma unroll
  for (int k = 0; k < kFunctorValues; k++) acc[k] = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const int c = corr[i];
    if (c < 0) continue;
    const float4 ps = output[i];
    const float4 pt = tgt[c];
    const float* M = maha9 + static_cast<size_t>(i) * 9;
    float px, py, pz;
    matvec_eigen(P.T, ps.x, ps.y, ps.z, px, py, pz);
    const float r0 = px - pt.x, r1 = py - pt.y, r2 = pz - pt.z;
    if (MODE != 1) {  // operator(), :241-274
      const float m0 = (M[0] * r0 + M[1] * r1) + M[2] * r2;
      const float m1 = (M[3] * r0 + M[4] * r1) + M[5] * r2;
      const float m2 = (M[6] * r0 + M[7] * r1) + M[8] * r2;
      const float ret = (r0 * m0 + r2 * m2) + r1 * m1;  // [Eigen] 4-wide dot: (p0 + p2) + (p1 + p3)
      acc[0] += static_cast<double>(ret);
    }
    if (MODE != 0) {  // df / fdf, :277-368
      const double d0 = static_cast<double>(r0), d1 = static_cast<double>(r1), d2 = static_cast<double>(r2);
      const double t0 = (static_cast<double>(M[0]) * d0 + static_cast<double>(M[1]) * d1) + static_cast<double>(M[2]) * d2;
      const double t1 = (static_cast<double>(M[3]) * d0 + static_cast<double>(M[4]) * d1) + static_cast<double>(M[5]) * d2;
      const double t2 = (static_cast<double>(M[6]) * d0 + static_cast<double>(M[7]) * d1) + static_cast<double>(M[8]) * d2;
      if (MODE == 1) acc[0] += (d0 * t0 + d1 * t1) + d2 * t2;
      acc[1] += t0;
      acc[2] += t1;
      acc[3] += t2;
      const double sx = static_cast<double>(ps.x), sy = static_cast<double>(ps.y), sz = static_cast<double>(ps.z);
      acc[4] += sx * t0;
      acc[5] += sx * t1;
      acc[6] += sx * t2;
      acc[7] += sy * t0;
      acc[8] += sy * t1;
      acc[9] += sy * t2;
      acc[10] += sz * t0;
      acc[11] += sz * t1;
      acc[12] += sz * t2;
    }
    acc[13] += 1.0;
  }
  // block row -> ticket -> fixed-order sum by the last block -> tagged publication (as k_derivatives_fused)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const double tot = wave_fold<kFunctorValues>(acc);
  if ((lane & 1) == 0) lds[wave * 32 + fold_index(lane)] = tot;
  __syncthreads();
  if (wave == 0) {
    if (lane < kEvalStride) {
      double v = 0.0;
      if (lane < kFunctorValues) {
        v = lds[lane];
#pragma unroll
        for (int w = 1; w < kWaves; w++) v += lds[w * 32 + lane];
      }
      __hip_atomic_store(partials + static_cast<size_t>(blockIdx.x) * kEvalStride + lane, v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (ticket == gridDim.x - 1) ? 1 : 0;
    }
  }
  __syncthreads();
  if (!s_last) return;
  const int k = threadIdx.x % kEvalStride, part = threadIdx.x / kEvalStride;
  lds2[part * kEvalStride + k] = sum_rows_fixed<kParts>(partials, gridDim.x, threadIdx.x);
  __syncthreads();
  if (threadIdx.x < kEvalStride) {
    double t = 0.0;
#pragma unroll
    for (int p = 0; p < kParts; p++) t += lds2[p * kEvalStride + threadIdx.x];
    lds[threadIdx.x] = t;
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  publish_row_tagged(out_row, lds, threadIdx.x, seq);
}

// ---------------------------------------------------------------------------
// Persistent objective server: ONE launch per BFGS run (the ~50 objective / gradient evaluations between two
// correspondence steps) instead of one launch per evaluation -- the protocol of the NDT evaluation server
// (ndt_latency.hip) with the direct mailbox: the host writes a 256-byte command (32 self-validating words: T[12], the
// mode) through the BAR into fine-grained device memory, every block reads it there, evaluates its slice of the
// correspondences, stores its partial row, takes a ticket on one of 8 shard counters (shard = part of k_functor's
// fixed-order sum), and a shard's last arriver publishes the part sum as a tagged row into pinned host memory; the host
// adds the 8 parts in order -- the same additions in the same order as k_functor's last block.
// Liveness: every spin is bounded by s_memrealtime budgets; block 0 tells the host when the server gives up.
// gridDim.x must not exceed the number of co-resident blocks (the launcher keeps it small).
// ---------------------------------------------------------------------------
constexpr int kGicpCmdWords = 32;
constexpr int kGicpCmdExit = 0x7fffffff;
struct GicpMailbox {  // layout of ndt_latency.hip's ServerMailbox (the host fills it with ndt::server_post)
  unsigned long long cmd[kGicpCmdWords];
  unsigned long long dead;
  unsigned long long pad[15];
};

__global__ __launch_bounds__(kBlock) void k_gicp_server(const float4* __restrict__ output, int n, const float4* __restrict__ tgt,
                                                        const int* __restrict__ corr, const float* __restrict__ maha9,
                                                        GicpMailbox* mb, double* __restrict__ partials,
                                                        unsigned* __restrict__ counter, double* __restrict__ out_rows,
                                                        unsigned long long first_seq, unsigned long long idle_ticks) {
#pragma clang fp contract(off)
  constexpr int kWaves = kBlock / kWave, kParts = kBlock / kEvalStride;  // 8 parts, as in k_functor
  static_assert(kParts == kGicpServerParts, "the host adds kGicpServerParts part sums");
  __shared__ double lds[kWaves * 32];
  __shared__ float sT[12];
  __shared__ int s_mode;
  unsigned long long expect = first_seq;
  for (;;) {
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (wave == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      const unsigned long long patience = (blockIdx.x == 0) ? idle_ticks : 4 * idle_ticks;
      unsigned long long w = 0;
      bool got = false;
      for (;;) {
        if (lane < kGicpCmdWords) w = __hip_atomic_load(&mb->cmd[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (lane == kGicpCmdWords) w = __hip_atomic_load(&mb->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__ballot(lane >= kGicpCmdWords || static_cast<unsigned>(w) == static_cast<unsigned>(expect)) == ~0ull) { got = true; break; }
        if (__ballot(lane == kGicpCmdWords && w == expect) != 0) break;  // block 0 gave up on this very command: leave with it
        if (__builtin_amdgcn_s_memrealtime() - t0 > patience) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (!got && blockIdx.x == 0 && lane == 0) __hip_atomic_store(&mb->dead, expect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      int mode = kGicpCmdExit;
      if (got) {
        const unsigned payload = static_cast<unsigned>(w >> 32);
        mode = static_cast<int>(__shfl(payload, 12, kWave));
        if (lane < 12) sT[lane] = __int_as_float(static_cast<int>(payload));
      }
      if (lane == 0) s_mode = mode;
    }
    __syncthreads();
    const int mode = s_mode;
    if (mode < 0 || mode > 3) return;  // EXIT, time-out or garbage: the whole block leaves together (0 f, 1 df, 2 fdf, 3 f + df)
    double acc[kFunctorValues];
#pragma unroll
    for (int k = 0; k < kFunctorValues; k++) acc[k] = 0.0;
    for (int i = blockIdx.x * kBlock + tid; i < n; i += gridDim.x * kBlock) {
      const int c = corr[i];
      if (c < 0) continue;
      const float4 ps = output[i];
      const float4 pt = tgt[c];
      const float* M = maha9 + static_cast<size_t>(i) * 9;
      float px, py, pz;
      matvec_eigen(sT, ps.x, ps.y, ps.z, px, py, pz);
      const float r0 = px - pt.x, r1 = py - pt.y, r2 = pz - pt.z;
      if (mode == 0 || mode == 3) {  // operator(), :241-274
        const float m0 = (M[0] * r0 + M[1] * r1) + M[2] * r2;
        const float m1 = (M[3] * r0 + M[4] * r1) + M[5] * r2;
        const float m2 = (M[6] * r0 + M[7] * r1) + M[8] * r2;
        const float ret = (r0 * m0 + r2 * m2) + r1 * m1;
        acc[0] += static_cast<double>(ret);
      }
      if (mode != 0) {  // df / fdf, :277-368
        const double d0 = static_cast<double>(r0), d1 = static_cast<double>(r1), d2 = static_cast<double>(r2);
        const double t0 = (static_cast<double>(M[0]) * d0 + static_cast<double>(M[1]) * d1) + static_cast<double>(M[2]) * d2;
        const double t1 = (static_cast<double>(M[3]) * d0 + static_cast<double>(M[4]) * d1) + static_cast<double>(M[5]) * d2;
        const double t2 = (static_cast<double>(M[6]) * d0 + static_cast<double>(M[7]) * d1) + static_cast<double>(M[8]) * d2;
        if (mode != 3) acc[0] += (d0 * t0 + d1 * t1) + d2 * t2;
        acc[1] += t0;
        acc[2] += t1;
        acc[3] += t2;
        const double sx = static_cast<double>(ps.x), sy = static_cast<double>(ps.y), sz = static_cast<double>(ps.z);
        acc[4] += sx * t0;
        acc[5] += sx * t1;
        acc[6] += sx * t2;
        acc[7] += sy * t0;
        acc[8] += sy * t1;
        acc[9] += sy * t2;
        acc[10] += sz * t0;
        acc[11] += sz * t1;
        acc[12] += sz * t2;
      }
      acc[13] += 1.0;
    }
    const double tot = wave_fold<kFunctorValues>(acc);
    if ((lane & 1) == 0) lds[wave * 32 + fold_index(lane)] = tot;
    __syncthreads();
    if (wave == 0) {
      if (lane < kEvalStride) {
        double v = 0.0;
        if (lane < kFunctorValues) {
          v = lds[lane];
#pragma unroll
          for (int w = 1; w < kWaves; w++) v += lds[w * 32 + lane];
        }
        __hip_atomic_store(partials + static_cast<size_t>(blockIdx.x) * kEvalStride + lane, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the ticket's answer, the shard's part sum and its publication stay inside this wave (as in k_eval_server): no LDS
      // stage, no block barrier -- the other waves wait at the round's last barrier
      const unsigned round = static_cast<unsigned>(expect - first_seq);
      const unsigned shard = blockIdx.x % static_cast<unsigned>(kParts);
      const unsigned in_shard = (gridDim.x + static_cast<unsigned>(kParts) - 1u - shard) / static_cast<unsigned>(kParts);
      unsigned t1 = 0u;
      if (lane == 0) t1 = __hip_atomic_fetch_add(counter + 32u * shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_readfirstlane(t1) == (round + 1u) * in_shard - 1u) {
        const double part = (lane < kEvalStride) ? sum_rows_fixed<kParts>(partials, gridDim.x, static_cast<int>(shard) * kEvalStride + lane) : 0.0;
        publish_lanes_tagged(out_rows + static_cast<size_t>(shard) * kPubWords, part, lane, expect);
      }
    }
    __syncthreads();  // s_mode / lds are rewritten by the next round
    expect++;
  }
}

}  // namespace

hipError_t launch_knn_covariances(const PointIndex& ix, int k, double gicp_epsilon, double* cov6, int* nn_idx, float* nn_d2,
                                  hipStream_t stream) {
  if (k < 1 || k > kMaxK) return hipErrorInvalidValue;
  constexpr int kQueriesPerBlock = kKnnBlock / kTeam;
  const int blocks = max(1, min(65536, (ix.n + kQueriesPerBlock - 1) / kQueriesPerBlock));
  const size_t lds = static_cast<size_t>(k) * kQueriesPerBlock * 12;  // per team: k distances, k positions, k ordered indices
  hipLaunchKernelGGL(k_knn_covariances, dim3(blocks), dim3(kKnnBlock), lds, stream, ix, k, gicp_epsilon, cov6, nn_idx, nn_d2);
  return hipGetLastError();
}

hipError_t launch_correspond(const float4* output, int n, const float* T12, const Rot3d& R, const PointIndex& tgt,
                             const double* cov_src6, const double* cov_tgt6, double dist_threshold, int* corr, float* maha9,
                             hipStream_t stream) {
  EvalParams P{};
  for (int i = 0; i < 12; i++) P.T[i] = T12[i];
  constexpr int kQueriesPerBlock = kBlock / kTeam;
  const int blocks = max(1, min(32768, (n + kQueriesPerBlock - 1) / kQueriesPerBlock));
  hipLaunchKernelGGL(k_correspond, dim3(blocks), dim3(kBlock), 0, stream, output, n, P, R, tgt, cov_src6, cov_tgt6,
                     dist_threshold, corr, maha9);
  return hipGetLastError();
}

int functor_blocks(int n) { return max(1, min(kFunctorMaxBlocks, (n + kBlock - 1) / kBlock)); }

hipError_t launch_functor(int mode, const float4* output, int n, const float4* tgt, const int* corr, const float* maha9,
                          const float* T12, int n_blocks, double* partials, unsigned* counter, double* out_row,
                          unsigned long long seq, hipStream_t stream) {
  EvalParams P{};
  for (int i = 0; i < 12; i++) P.T[i] = T12[i];
  if (mode == 0)
    hipLaunchKernelGGL(k_functor<0>, dim3(n_blocks), dim3(kBlock), 0, stream, output, n, tgt, corr, maha9, P, partials, counter,
                       out_row, seq);
  else if (mode == 3)
    hipLaunchKernelGGL(k_functor<2>, dim3(n_blocks), dim3(kBlock), 0, stream, output, n, tgt, corr, maha9, P, partials, counter,
                       out_row, seq);
  else
    hipLaunchKernelGGL(k_functor<1>, dim3(n_blocks), dim3(kBlock), 0, stream, output, n, tgt, corr, maha9, P, partials, counter,
                       out_row, seq);
  return hipGetLastError();
}

int server_blocks(int n) { return max(1, min(512, (n + kBlock - 1) / kBlock)); }

hipError_t launch_server(const float4* output, int n, const float4* tgt, const int* corr, const float* maha9, void* mailbox,
                         int n_blocks, double* partials, unsigned* counter, double* out_rows, unsigned long long first_seq,
                         unsigned long long idle_ticks, hipStream_t stream) {
  if (ndt::server_mailbox_bytes() != sizeof(GicpMailbox)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_gicp_server, dim3(n_blocks), dim3(kBlock), 0, stream, output, n, tgt, corr, maha9,
                     static_cast<GicpMailbox*>(mailbox), partials, counter, out_rows, first_seq, idle_ticks);
  return hipGetLastError();
}

}  // namespace gicp
