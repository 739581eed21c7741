// ndt_search.hpp -- exact nearest-neighbour search over a PointIndex (a cloud + the counting-sort voxel
// index K1 builds over it), shared by getFitnessScore (ndt_kernels.hip) and the GICP kernels
// (gicp_kernels.hip).  Device code in an anonymous namespace, like ndt_device.hpp.
//
// Search: the query's cell, then cubic shells of cells around it, until no unvisited shell can hold a
// better point -- exact; ties are resolved by the callers on (distance, index).
#pragma once
#include "ndt_device.hpp"

namespace ndt {
namespace {

constexpr int kKnnMinRing = 3;  // shells every query may try (see max_shells) before one scan over all points

// The query's cell (clamped into the grid) and its MARGIN: how far the query is from the nearest face of that cell
// (0 for a query outside the grid).  After shell r every unvisited point is at least r * leaf + margin away -- with
// the margin a query in a crowded cell can stop after its own cell (r = 0) instead of always walking 27.
__device__ __forceinline__ void query_cell(const GridGeom& g, float x, float y, float z, int& ci, int& cj, int& ck, float& margin) {
  int ri, rj, rk;
  search_ijk(g, x, y, z, ri, rj, rk);
  ci = max(g.min_b[0], min(g.max_b[0], ri));
  cj = max(g.min_b[1], min(g.max_b[1], rj));
  ck = max(g.min_b[2], min(g.max_b[2], rk));
  margin = 0.0f;
  if (ci == ri && cj == rj && ck == rk) {
    const float fx = x - static_cast<float>(ri) * g.leaf[0], fy = y - static_cast<float>(rj) * g.leaf[1], fz = z - static_cast<float>(rk) * g.leaf[2];
    margin = fminf(fminf(fminf(fx, g.leaf[0] - fx), fminf(fy, g.leaf[1] - fy)), fminf(fz, g.leaf[2] - fz));
    margin = fmaxf(margin, 0.0f);
  }
  ci -= g.min_b[0];  // nearest grid cell when outside the bounding box
  cj -= g.min_b[1];
  ck -= g.min_b[2];
}

// ---------------------------------------------------------------------------
// Team search.  A query is worked on by kTeam = 8 adjacent lanes: they look up the same cell and
// stride over its points, each lane keeping its own best / its own sorted candidate list, and the
// team combines them with three xor-shuffles.  With a thread per query the few 10^4 queries of a
// down-sampled scan leave the chip almost empty and every query is one long chain of load
// latencies (measured on the reference pair: 2 ms per correspondence step, 4 ms per covariance
// pass); teams cut the chain eightfold and give the machine eight times the waves.
// ---------------------------------------------------------------------------
constexpr int kTeam = 8;

// squared distances from (qx,qy,qz) to this lane's share of the `count` cell-ordered points starting at
// `first` (positions sub, sub + 8, ...), two loads in flight.  visit(d, position in the cell order, valid) is
// called the same number of times by every lane of the team (valid = false past the end), so that a visitor may
// use team-wide operations.
template <int TEAM = kTeam, class F>
__device__ __forceinline__ void scan_run(const float4* __restrict__ sp, unsigned first, int count, int sub, float qx, float qy,
                                         float qz, F&& visit) {
  for (int base = 0; base < count; base += 2 * TEAM) {
    const int p0 = base + sub, p1 = p0 + TEAM;
    const bool v0 = p0 < count, v1 = p1 < count;
    // clamped, not predicated: both loads issue together (count >= 1 here)
    const float4 a = sp[first + min(p0, count - 1)], b = sp[first + min(p1, count - 1)];
    const float da = dist2_f32(qx, qy, qz, a.x, a.y, a.z), db = dist2_f32(qx, qy, qz, b.x, b.y, b.z);
    visit(da, first + p0, v0);
    visit(db, first + p1, v1);
  }
}

// the exhaustive scan of a query without near neighbours: this lane's share of all points, four loads in flight;
// visit(d, position, point, valid), uniform across the team like scan_run
template <class F>
__device__ __forceinline__ void scan_all(const float4* __restrict__ sp, int count, int sub, float qx, float qy, float qz, F&& visit) {
  for (int base = 0; base < count; base += 4 * kTeam) {
    const int p0 = base + sub, p1 = p0 + kTeam, p2 = p1 + kTeam, p3 = p2 + kTeam;
    const bool v0 = p0 < count, v1 = p1 < count, v2 = p2 < count, v3 = p3 < count;
    const float4 a = sp[min(p0, count - 1)], b = sp[min(p1, count - 1)], c = sp[min(p2, count - 1)], d = sp[min(p3, count - 1)];
    const float da = dist2_f32(qx, qy, qz, a.x, a.y, a.z), db = dist2_f32(qx, qy, qz, b.x, b.y, b.z);
    const float dc = dist2_f32(qx, qy, qz, c.x, c.y, c.z), dd = dist2_f32(qx, qy, qz, d.x, d.y, d.z);
    visit(da, static_cast<unsigned>(p0), a, v0);
    visit(db, static_cast<unsigned>(p1), b, v1);
    visit(dc, static_cast<unsigned>(p2), c, v2);
    visit(dd, static_cast<unsigned>(p3), d, v3);
  }
}

// the cell a point was binned into when the index was built (ndt_kernels.hip build_cell: floor(x * inv_leaf) - float(min_b),
// f32, product rounded before floor) -- tells whether the shells already walked have covered it
__device__ __forceinline__ bool in_walked_cube(const GridGeom& g, const float4& p, int ci, int cj, int ck, int r_done) {
#pragma clang fp contract(off)
  const float fx = p.x * g.inv_leaf[0], fy = p.y * g.inv_leaf[1], fz = p.z * g.inv_leaf[2];
  const int i0 = static_cast<int>(floorf(fx) - static_cast<float>(g.min_b[0]));
  const int i1 = static_cast<int>(floorf(fy) - static_cast<float>(g.min_b[1]));
  const int i2 = static_cast<int>(floorf(fz) - static_cast<float>(g.min_b[2]));
  return abs(i0 - ci) <= r_done && abs(i1 - cj) <= r_done && abs(i2 - ck) <= r_done;
}

// lexicographic minimum of (d, idx) over the 8 lanes of a team
__device__ __forceinline__ void team_min(float& d, int& idx) {
#pragma unroll
  for (int off = 1; off < kTeam; off <<= 1) {
    const float od = __shfl_xor(d, off, kWave);
    const int oi = __shfl_xor(idx, off, kWave);
    if (od < d || (od == d && oi < idx)) {
      d = od;
      idx = oi;
    }
  }
}
__device__ __forceinline__ int team_sum(int v) {
#pragma unroll
  for (int off = 1; off < kTeam; off <<= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Shell r of cells around (ci, cj, ck), worked on by a team: the (2r+1)^2 rows of the shell are dealt out
// to the eight lanes, each lane probes one cell of its row per step (face rows: every x; interior rows:
// only x = -r and x = +r), and every occupied cell any lane finds is then scanned by the whole team.
// consider(d, position) as in scan_run.  All control flow is uniform within the team.
// bound2 (optional): a squared distance the caller can no longer use -- rows and cells that cannot hold a point
// nearer than that are not even looked up.  The distance from the query to a cell's box, less the index-rounding slack
// (a point may sit that far outside the cell it was binned into), bounds every point of the cell from below; after the
// query's own cell has given a neighbour at ~ the point spacing, that leaves one to three of a shell's 26 cells.
__device__ __forceinline__ float axis_gap(float q, int cell_abs, float leaf, float slack) {
  const float lo = static_cast<float>(cell_abs) * leaf, hi = lo + leaf;
  return fmaxf(fmaxf(lo - q, q - hi) - slack, 0.0f);
}
template <int TEAM = kTeam, class F>
__device__ __forceinline__ void team_shell(const PointIndex& ix, int ci, int cj, int ck, int r, int sub, float qx, float qy,
                                           float qz, F&& consider, float bound2 = INFINITY) {
  // (TEAM = 8: the team search; TEAM = 64: a whole wave on one query -- the far queries of getFitnessScore, whose shells
  // have hundreds of rows)
  constexpr unsigned long long kTeamMask = TEAM >= 64 ? ~0ull : ((1ull << (TEAM & 63)) - 1ull);
  const GridGeom& g = ix.geom;
  const int w = 2 * r + 1, rows = w * w;
  const int team_base = (threadIdx.x & (kWave - 1)) & ~(TEAM - 1);
  const bool prune = bound2 < INFINITY;
  for (int row0 = 0; row0 < rows; row0 += TEAM) {
    const int row = row0 + sub;
    const int dz = row / w - r, dy = row % w - r;
    const int z = ck + dz, y = cj + dy;
    bool row_ok = row < rows && z >= 0 && z < g.div_b[2] && y >= 0 && y < g.div_b[1];
    float row_d2 = 0.0f;
    if (prune && row_ok) {
      const float gy = axis_gap(qy, y + g.min_b[1], g.leaf[1], ix.slack), gz = axis_gap(qz, z + g.min_b[2], g.leaf[2], ix.slack);
      row_d2 = gy * gy + gz * gz;
      row_ok = row_d2 <= bound2;
    }
    if (row_ok) row_ok = ix.row_any[y + z * g.div_b[1]] != 0;  // rows without a single occupied cell cost one load
    if (((__ballot(row_ok) >> team_base) & kTeamMask) == 0) continue;
    const bool face = (dz == -r || dz == r || dy == -r || dy == r);  // r == 0: the single row is a face row
    const int nx = face ? w : 2;
    int steps = row_ok ? nx : 0;  // the longest occupied row of this batch sets the number of steps (interior rows: 2 cells)
#pragma unroll
    for (int off = 1; off < TEAM; off <<= 1) steps = max(steps, __shfl_xor(steps, off, kWave));
    for (int t = 0; t < steps; t += 2) {  // two cells of the row per step: both table loads are in flight together
      uint2 range[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int tt = t + u;
        if (row_ok && tt < nx) {
          const int x = ci + (face ? tt - r : (tt == 0 ? -r : r));
          bool cell_ok = x >= 0 && x < g.div_b[0];
          if (prune && cell_ok) {
            const float gx = axis_gap(qx, x + g.min_b[0], g.leaf[0], ix.slack);
            cell_ok = row_d2 + gx * gx <= bound2;
          }
          if (cell_ok) range[u] = ix.cell_range[x * g.mul[0] + y * g.mul[1] + z * g.mul[2]];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const unsigned first = range[u].x;
        const int count = static_cast<int>(range[u].y);
        unsigned long long found = (__ballot(count > 0) >> team_base) & kTeamMask;
        while (found) {
          const int owner = __builtin_ctzll(found);
          found &= found - 1;
          const unsigned fs = __shfl(first, team_base + owner, kWave);
          const int fc = __shfl(count, team_base + owner, kWave);
          scan_run<TEAM>(ix.sorted_pts, fs, fc, sub, qx, qy, qz, consider);
        }
      }
    }
  }
}

// The scan over all points for a single nearest neighbour, by the WHOLE wave: the few isolated queries that need it
// set a search kernel's time when one team of 8 lanes walks the cloud alone.  Every lane of the wave must call this
// with the same query; returns the smallest (distance, point index) in all lanes.
__device__ __forceinline__ void wave_nearest(const PointIndex& ix, float qx, float qy, float qz, float& out_d, int& out_idx) {
  const int lane = threadIdx.x & (kWave - 1);
  float best = INFINITY;
  int best_p = -1;
  auto take = [&](float d, int pos, bool ok) {
    if (!ok || d > best) return;
    if (d < best || ix.sorted_idx[pos] < ix.sorted_idx[best_p]) {
      best = d;
      best_p = pos;
    }
  };
  const int count = ix.n_sorted;
  for (int base = 0; base < count; base += 4 * kWave) {
    const int p0 = base + lane, p1 = p0 + kWave, p2 = p1 + kWave, p3 = p2 + kWave;
    const float4 a = ix.sorted_pts[min(p0, count - 1)], b = ix.sorted_pts[min(p1, count - 1)];
    const float4 c = ix.sorted_pts[min(p2, count - 1)], d = ix.sorted_pts[min(p3, count - 1)];
    take(dist2_f32(qx, qy, qz, a.x, a.y, a.z), p0, p0 < count);
    take(dist2_f32(qx, qy, qz, b.x, b.y, b.z), p1, p1 < count);
    take(dist2_f32(qx, qy, qz, c.x, c.y, c.z), p2, p2 < count);
    take(dist2_f32(qx, qy, qz, d.x, d.y, d.z), p3, p3 < count);
  }
  float d = best;
  int idx = best_p >= 0 ? ix.sorted_idx[best_p] : 0x7fffffff;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const float od = __shfl_xor(d, off, kWave);
    const int oi = __shfl_xor(idx, off, kWave);
    if (od < d || (od == d && oi < idx)) {
      d = od;
      idx = oi;
    }
  }
  out_d = d;
  out_idx = idx;
}

// Shells tried before a query falls back to one scan over all points.  Shell r costs the team up to
// (2r+1)^3 / 8 probe steps, the shells up to r about (2r+1)^4 / 64 together; the scan costs n / 32 steps
// (8 lanes, four loads in flight): stop walking shells where the two meet.
__device__ __forceinline__ int max_shells(const PointIndex& ix, int r_lim) {
  const int by_cost = (static_cast<int>(sqrtf(sqrtf(2.0f * static_cast<float>(ix.n_sorted)))) - 1) / 2;
  return min(r_lim, max(kKnnMinRing, by_cost));
}

}  // namespace
}  // namespace ndt
