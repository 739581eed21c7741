// ndt_kernels.hpp -- device-side data layout shared by the HIP kernels and the
// C-ABI glue of the MI355X NDT core (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace ndt {

// One valid target voxel, exactly one 64-byte sector (half a 128-B L2 line):
//   mean  : 3 x f64  -- the reference subtracts the f64 mean from the f32 point
//                       in f64 and only then rounds to f32 (ndt_omp_impl.hpp:262,492)
//   icov  : the six entries of Sigma^-1 (symmetric) as the f32 the reference casts to per use (:494), in the order
//           c00 c01 c02 c12 c11 c22: with the means the FIRST 48 BYTES of the record -- three 16-byte loads per neighbour
//           (every gather instruction of every wave goes through the CU's one vector memory path).  The packed f32 math
//           multiplies by four register pairs: (c00,c01) (c02,c12) (c11,c22) are adjacent in the loads, (c01,c11) is
//           assembled with a move -- or, for the one kernel that is better off with a fourth load than with ten more
//           registers (the launch path's one-launch kernel: 126 VGPRs, four waves per SIMD), read from the copy at the
//           end of the sector.
//   n     : point count.  (The voxel centroid and the f64 inverse covariance live in the side sector, VoxelSide.)
struct alignas(64) VoxelRec {
  double mean[3];
  float c[6];
  int n;
  int pad;
  float c01c11[2];
};
static_assert(sizeof(VoxelRec) == 64, "VoxelRec must be one 64-B sector");
// What the f32 evaluation does not read, one 64-B sector per record again: the voxel centroid (voxel_centroids_ entry,
// KDTREE search) and the inverse covariance in f64 -- computeHessian (ndt_omp_impl.hpp:540-645) and calculateScore
// (:935-983) multiply with the f64 icov_ of the leaf, not with its f32 rounding.  icov: c00 c01 c02 c11 c12 c22.
struct alignas(64) VoxelSide {
  float cx, cy, cz, pad;
  double icov[6];
};
static_assert(sizeof(VoxelSide) == 64, "VoxelSide must be one 64-B sector");

// VoxelGridCovariance geometry (voxel_grid_covariance_omp_impl.hpp:87-103).
struct GridGeom {
  float leaf[3];
  float inv_leaf[3];
  int min_b[3];
  int max_b[3];
  int div_b[3];
  int mul[3];
  long long n_cells;
  // the voxel look-up table of the evaluation kernels is laid out over the bounding box PLUS a border of
  // kLutBorder empty cells on every side: a point within one cell of the box (near_grid) probes its 7 / 26 / 27
  // neighbour cells without a bounds test per probe (the reference's test, _impl.hpp:382-392, is what the border
  // encodes).  pmul: strides of that padded table; lut_cells: its size.
  int pmul[3];
  long long lut_cells;
  int pow2;  // every leaf size is a power of two: floor(x / leaf) == floor(x * inv_leaf) exactly
  // sparse grids: the look-up table is an open-addressing hash keyed by the reference's linear voxel index
  // (hash_bits > 0: 2^hash_bits slots of (key, entry), key -1 = free); lut_cells / pmul are then unused
  int hash_bits;
};
constexpr int kLutBorder = 2;
inline void set_padded_lut(GridGeom& g) {
  g.pmul[0] = 1;
  g.pmul[1] = g.div_b[0] + 2 * kLutBorder;
  g.pmul[2] = g.pmul[1] * (g.div_b[1] + 2 * kLutBorder);
  g.lut_cells = static_cast<long long>(g.pmul[2]) * (g.div_b[2] + 2 * kLutBorder);
  g.hash_bits = 0;
  g.pow2 = 1;
  for (int k = 0; k < 3; k++) {
    int e = 0;
    const float m = std::frexp(g.leaf[k], &e);
    if (m != 0.5f) g.pow2 = 0;
  }
}

// LUT entry: record index (>= 0) of a valid voxel; kLutEmpty; or lut_rejected(r) (<= -2) for a voxel that
// reached min_points_per_voxel but was rejected (nr_points = -1, _impl.hpp:337-341,360-364): the DIRECT
// searches skip it, the KDTREE search still sees its record (trap 7).
constexpr int kLutEmpty = -1;
__host__ __device__ inline int lut_rejected(int r) { return -(r + 2); }

__host__ __device__ inline unsigned hash_slot(int key, int bits) { return (static_cast<unsigned>(key) * 0x9E3779B1u) >> (32 - bits); }

struct GridView {
  const int* lut;            // padded table, g.lut_cells entries; sparse grids (g.hash_bits > 0): the hash table, int2 slots
  const VoxelRec* recs;      // one record per voxel with >= min_points_per_voxel points
  const VoxelSide* centroids;  // per record: voxel centroid (KDTREE search) + f64 inverse covariance (f64 Hessian, calculateScore)
  GridGeom g;
};

// A cloud together with the counting-sort voxel index K1 built over it (ndt_kernels.hip): the exact
// nearest-neighbour searches walk cubic shells of cells around the query.
struct PointIndex {
  const float4* pts = nullptr;  // caller's order
  int n = 0;
  ndt::GridGeom geom{};
  const uint2* cell_range = nullptr;     // n_cells: (first position, count) of the cell's points in the cell order; count 0 = empty
  const int* row_any = nullptr;          // div_y * div_z: 1 where the x-row (y, z) has an occupied cell
  const int* sorted_idx = nullptr;  // point indices grouped by cell
  const float4* sorted_pts = nullptr;  // the points in that order (launch_gather_points)
  int n_sorted = 0;
  float slack = 0.f;  // build-time vs search-time cell index rounding (SURVEY 8a trap 2)
};

// Per-evaluation constants of computeDerivatives (ndt_omp_impl.hpp:179-285).
struct EvalParams {
  float T[12];      // row-major 3x4 f32 transform applied to the source
  float j[8][3];    // j_ang  (:338-346)
  float h[15][3];   // h_ang  (:373-393), row 6 with +sy
  double d1;        // gauss_d1_ (used as double, :501,510)
  float d2;         // gauss_d2_ cast to float (:496)
  int pad;          // KDTREE: bits of the f32 squared search radius
};

// computeHessian's all-f64 constants (:540-645).
struct Hess64Params {
  float T[12];
  double jd[8][3];
  double hd[15][3];
  double d1, d2;
  double r2;   // KDTREE: squared search radius (f32 value)
};

// Batched launches: one descriptor per scan.
struct ScanDesc {
  int offset;  // first point of the scan in the concatenated source
  int count;
  int kind;    // ndt::EvalKind of this step (EVAL_NONE = skip)
  int pad;     // batch: partial rows written for this scan in this step
  EvalParams P;
  Hess64Params P64;
};

// Packed result row: score, g[6], H upper triangle row-major [21], n_neighbors,
// 3 spare  (NDT_EVAL_STRIDE doubles).
constexpr int kEvalStride = 32;
constexpr int kNumAcc = 29;

// --- launchers (ndt_kernels.hip) -------------------------------------------
// All launch on `stream` and return the first HIP error.
hipError_t launch_repack(const void* d_src, size_t n, size_t stride_bytes, float4* d_dst, hipStream_t stream);

// K1 in its bucket form (ndt_kernels.hip): the voxel index space cut into n_buckets runs of cells_per_bucket = 2^shift
// consecutive cells; pts_per_block points per block of the two point passes.
struct GridBuildPlan {
  int shift, n_buckets, cells_per_bucket, pts_per_block, n_blocks;
};
// false: the grid is outside the bucket form's range (more than 8192 x 4096 cells): the general path builds it
bool grid_build_plan(long long n_cells, int n_points, GridBuildPlan& plan);
constexpr int kK1MaxBuckets = 8192;
struct GridBuildScratch {
  unsigned* cntmat;        // [n_blocks x n_buckets] points per (block of points, bucket), then [n_buckets] bucket sizes
  unsigned* bucket_base;   // [n_buckets + 1] bases + [n_buckets] valid voxels per bucket   (kept with the grid: the leaf pass needs both)
  float4* bpts;            // [n] points in bucket order, w = point index   (kept with the grid until the leaf pass)
  unsigned* order;         // [5 n] per-point scratch of voxels too crowded for LDS
  unsigned long long* stamps;  // development aid (NDT_K1_STAMPS): [n_buckets x 8] phase clocks of k1_finalize, or null
  bool index_form;             // experiment (NDT_K1_INDEX=1): bpts holds 4-byte point indices, k1_finalize gathers from the cloud
};
// counts: device [4] = {points binned, occupied voxels, candidate voxels, valid voxels}.  The build fills [0] and [3];
// [1], [2] and the leaf arrays come from launch_grid_leaves (on demand).  Record slots: n / min_pts + 1.
// pcl::VoxelGrid's centroid filter on the same front end (ndt_grid_kernels.hip "N1 / N2 on the bucket front end"): out receives
// one centroid per occupied voxel in ascending voxel index; counts[1] = how many.  st_cell / st_cent: n entries each; bitmap_words
// / wprefix: filter_buckets_bitmap_words(n_cells) words each.
bool filter_buckets_plan(long long n_cells, int n_points, GridBuildPlan& plan);
size_t filter_buckets_bitmap_words(long long n_cells);
hipError_t launch_filter_buckets(const float4* pts, int n, int dense, const GridGeom& g, const GridBuildPlan& plan, const GridBuildScratch& scratch,
                                 int* st_cell, float4* st_cent, unsigned* bitmap_words, unsigned* wprefix, unsigned* counts, float4* out,
                                 hipStream_t stream);
// the same grid from ONE launch, for small clouds (every block scans the whole cloud and finishes its own bucket); scratch.cntmat unused
bool grid_build_small_applies(int n_points, const GridBuildPlan& plan);
hipError_t launch_grid_build_small(const float4* pts, int n, int dense, const GridGeom& g, const GridBuildPlan& plan, int min_pts,
                                   double eig_ratio, const GridBuildScratch& scratch, int* sorted_idx, VoxelRec* recs, VoxelSide* centroids,
                                   int* lut, unsigned* counts, hipStream_t stream);
hipError_t launch_grid_build_buckets(const float4* pts, int n, int dense, const GridGeom& g, const GridBuildPlan& plan, int min_pts,
                                     double eig_ratio, const GridBuildScratch& scratch, int* sorted_idx, VoxelRec* recs, VoxelSide* centroids,
                                     int* lut, unsigned* counts, hipStream_t stream);
hipError_t launch_grid_leaves(const GridGeom& g, const GridBuildPlan& plan, int min_pts, const float4* bpts, const unsigned* bucket_base,
                              unsigned* scratch /* 4 n_buckets + 4 words */, int* leaf_cell, unsigned* leaf_start, int* leaf_count,
                              int* leaf_rec, unsigned* counts, const int* lut, hipStream_t stream, const float4* cloud = nullptr);  // cloud: the index form's points
// records of a bucket-form build -> dense, in ascending cell order (table entries rewritten); tile_sums: record_compaction_tiles words
size_t record_compaction_tiles(long long lut_cells);
hipError_t launch_compact_records(int* lut, long long lut_cells, const VoxelRec* recs_in, const VoxelSide* cent_in, VoxelRec* recs_out,
                                  VoxelSide* cent_out, unsigned* tile_sums, hipStream_t stream);

// n dense records from page-locked host memory (read by the kernel itself) into HBM
hipError_t launch_copy_records(const float4* src_host_pinned, float4* dst, int n, hipStream_t stream);
// repack + bounding boxes (block rows of 12 floats: non-NaN min/max xyz, finite-only min/max xyz); 16-byte records on a
// 16-byte boundary take a float4 path, and with d_dst == nullptr only the boxes are computed (cloud used where it lies)
hipError_t launch_repack_bbox(const void* d_src, size_t n, size_t stride_bytes, float4* d_dst, float* d_block_minmax,
                              int n_blocks, hipStream_t stream, unsigned tag = 0, const unsigned* n_dev = nullptr);  // n_dev: the count, still on the device
hipError_t launch_bbox(const float4* pts, int n, int dense, float* d_block_minmax, int n_blocks, hipStream_t stream);
hipError_t launch_count(const float4* pts, int n, int dense, const GridGeom& g, int* d_key, unsigned* d_rank,
                        unsigned* d_cell_count, hipStream_t stream);
// Source ordering of a lock-step batch: every scan on a lattice of ITS OWN (pitch and box from its own points only), so
// that a scan's order -- and with it the order of its f64 sums -- does not depend on the batch around it.
struct ScanLattice {
  float inv_leaf;   // 1 / pitch
  int min_b[3];     // floor(min * inv_leaf) of the scan's own bounding box
  int mul1, mul2;   // strides of y and z (x: 1)
  int n_cells;      // cells of the scan's box; 0: no finite point
  long long base;   // first composite cell of the scan in this pass's counter array
};
// bounding boxes per scan, order-preserving int encoding of the floats: out[scan][0..2] = min xyz, [3..5] = max xyz
// (the caller fills out with INT_MAX / INT_MIN first); scan_bbox_decode turns an encoded value back into the float
hipError_t launch_scan_bboxes(const float4* pts, const int* d_scan_off, int n_scans, int max_scan_points, int* d_out, hipStream_t stream);
float scan_bbox_decode(int v);
hipError_t launch_count_batch(const float4* pts, const int* d_scan_off, int n_scans, int max_scan_points, const ScanLattice* d_lat,
                              int* d_key, unsigned* d_rank, unsigned* d_cell_count, hipStream_t stream);
// out[k] = cell_count[bases[k]] (k < n): where every scan's ordered segment starts, after the exclusive scan
hipError_t launch_pick(const unsigned* cell_count, const long long* d_bases, unsigned* d_out, int n, hipStream_t stream);
hipError_t launch_scan_reduce(const unsigned* d_cell_count, long long n_cells, int min_pts, unsigned* d_block_sums,
                              int n_tiles, hipStream_t stream);
hipError_t launch_scan_blocks(unsigned* d_block_sums, int n_tiles, unsigned* d_totals, hipStream_t stream);
hipError_t launch_scan_apply(unsigned* d_cell_count_to_cursor, long long n_cells, int min_pts,
                             const unsigned* d_block_sums, int n_tiles, int* d_leaf_cell,
                             unsigned* d_leaf_start, int* d_leaf_count, int* d_leaf_rec, hipStream_t stream);
hipError_t launch_scatter(const int* d_key, const unsigned* d_rank, int n, const unsigned* d_cell_start, int* d_sorted_idx,
                          hipStream_t stream);

// Sparse voxel index (ndt_sparse.hip): leaf arrays in ascending voxel order from a stable sort of (voxel, point) pairs.
// counts: device [5] = {points binned, occupied voxels, candidates, -, -} (zeroed here; the finalize pass adds [3], [4]).
// keys_a / keys_b / vals_a / flags / ord: n words of scratch each; temp: sparse_index_temp_bytes(n).
size_t sparse_index_temp_bytes(int n);
hipError_t launch_sparse_index(const float4* pts, int n, int dense, const GridGeom& g, int min_pts, void* temp, size_t temp_bytes,
                               unsigned* keys_a, unsigned* keys_b, int* vals_a, unsigned* flags, unsigned* ord, int* leaf_cell,
                               unsigned* leaf_start, int* leaf_count, int* leaf_rec, int* sorted_idx, unsigned* counts,
                               hipStream_t stream);

struct FinalizeDump {  // optional per-leaf outputs for ndt_grid_dump
  int* nr_points;
  double* mean;
  double* cov;
  double* icov;
  double* evals;
};
hipError_t launch_finalize(const float4* pts, const int* d_leaf_cell, const unsigned* d_leaf_start,
                           const int* d_leaf_count, const int* d_leaf_rec, int n_leaves, int* d_sorted_idx,
                           int min_pts, double eig_ratio, VoxelRec* d_recs, VoxelSide* d_centroids, int* d_lut, const GridGeom& geom,
                           unsigned* d_n_valid, FinalizeDump dump, hipStream_t stream, const unsigned* d_totals = nullptr,
                           float4* d_big_pts = nullptr /* n points of scratch: enables the wave pre-sort of crowded leaves */);

// K2: derivatives.  search: NDT_DIRECT26/7/1.  Single scan: descs == nullptr, params by value,
// partials [n_blocks][kEvalStride].  Batch: grid.y walks active[0..n_active) -- the scans that want
// THIS kind of evaluation in this step -- with n_blocks blocks each; partial rows of scan s live at
// [s][max_blocks][kEvalStride] and descs[s].pad tells the reduce how many rows were written.
hipError_t launch_derivatives(const float4* src, int n, const GridView& gv, const EvalParams& P, int search,
                              bool want_hessian, const ScanDesc* descs, const int* active, int n_active, int max_blocks,
                              int n_blocks, double* partials, hipStream_t stream);
// Single launch: derivatives + fixed-order final sum by the last-arriving block + publication of
// the packed row and `seq` into pinned host memory (out_row).  counter: one zero-initialised u32.
// points per 512-thread block of the latency kernels: a block per CU of the handle's `cus` while 512 points per block allow
// that (64 at least: small scans use only the first lanes of a block); blocks of the one-launch kernel (two per CU at most
// on a CU partition)
int points_per_block(int n, int cus);
int fused_blocks(int n, int cus, bool partition);
hipError_t launch_derivatives_fused(const float4* src, int n, const GridView& gv, const EvalParams& P, int search,
                                    bool want_hessian, int n_blocks, int ppb, double* partials, unsigned* counter, double* out_row,
                                    unsigned long long seq, hipStream_t stream);
// Persistent evaluation server (one launch per align): see ndt_kernels.hip.
constexpr int kPublishSlots = 64;  // tagged publication row: value k as words 2k, 2k+1 = (half << 32) | seq32
constexpr int kServerParts = 16;   // the evaluation server publishes one row per PART of the fixed-order sum (rows b, b % 16 == part);
                                   // the host adds the parts in order (= server block size / kEvalStride)
constexpr int kServerCmdExit = 0x7fffffff;
constexpr int kServerCmdTransformExit = 4;  // write the aligned cloud with the command's transform, then exit
size_t server_mailbox_bytes();
void server_reset_mailbox(void* host_mailbox);
void server_post(void* host_mailbox, unsigned long long seq, int kind, const float* T12, const double* cos_sin6);
unsigned long long server_dead_word(const void* host_mailbox);
// diagnostic: the server's round driven from the device (k_selfdrive, ndt_latency.hip); server_post fills a host copy of the mailbox
hipError_t launch_selfdrive(const float4* src, int n, const GridView& gv, int search, void* dev_mailbox, int n_blocks, int ppb, double* partials,
                            unsigned* counter, double* parts, double* out_row, unsigned long long first_seq, int rounds, int with_body,
                            double gauss_d1, double gauss_d2, int param_pad, hipStream_t stream);
hipError_t launch_eval_server(const float4* src, int n, const GridView& gv, int search, void* host_mailbox,
                              void* dev_mailbox, int n_blocks, int ppb, double* partials, unsigned* counter, double* out_row,
                              unsigned long long first_seq, unsigned long long idle_ticks, double gauss_d1, double gauss_d2,
                              int param_pad, const float4* out_src, float4* out_dst, int out_n, hipStream_t stream,
                              unsigned long long* dbg, int direct, float4* out_host, unsigned* counter_next);
constexpr int kServerCounterWords = 32 * (1 + kServerParts);  // one set of the server's shard counters (a 128-B line each)
hipError_t launch_hessian64(const float4* src, int n, const GridView& gv, const Hess64Params& P, int search,
                            const ScanDesc* descs, const int* active, int n_active, int max_blocks, int n_blocks,
                            double* partials, hipStream_t stream);
// Sums the per-block partials in a fixed order: out[scan][kEvalStride].  seq != 0: `out` is pinned
// host memory polled by the host; slot kEvalStride-1 of each row then receives `seq` (u64) last.
hipError_t launch_reduce(const double* partials, int n_blocks, int n_scans, const ScanDesc* descs, double* out,
                         hipStream_t stream, unsigned long long seq = 0);
// n_rows x kEvalStride doubles of device memory -> pinned host rows, each followed by `seq` in slot kEvalStride-1
hipError_t launch_publish_rows(const double* d_rows, int n_rows, double* host_rows, unsigned long long seq, hipStream_t stream);
// all live scans of a lock-step batch step in one launch, kinds mixed (descs[scan].kind, .pad = rows written)
hipError_t launch_batch_step(const float4* src, const GridView& gv, int search, const ScanDesc* descs, const int* active,
                             int n_active, int max_blocks, int n_blocks, double* partials, hipStream_t stream);
// cell_range (pre-set to 0) and row_any (pre-set to 0, div_y * div_z entries) of a built grid
hipError_t launch_cell_ranges(const int* leaf_cell, const unsigned* leaf_start, const int* leaf_count, int n_leaves, uint2* cell_range,
                              int div_x, int* row_any, hipStream_t stream);
// [PCL] getFitnessScore: team search over the target's point index (ndt_search.hpp); partials [n_blocks][kEvalStride]
hipError_t launch_fitness(const float4* src, int n, const float* T12, const PointIndex& tgt, double max_range, int n_blocks,
                          double* partials, hipStream_t stream);
// pts in the cell order of an index: out[q] = pts[sorted_idx[q]] for q < *d_n_sorted
hipError_t launch_gather_points(const float4* pts, const int* sorted_idx, const unsigned* d_n_sorted, int n_max, float4* out,
                                hipStream_t stream);
hipError_t launch_transform(const float4* src, int n, const float* T12, float4* dst, hipStream_t stream, int dense = 1);
hipError_t launch_calc_score(const float4* cloud, int n, const GridView& gv, double d1, double d2, double d3,
                             int search, float r2, int n_blocks, double* partials, hipStream_t stream);

hipError_t launch_voxel_centroids(const float4* pts, const unsigned* leaf_start, const int* leaf_count, int n_leaves,
                                  int* sorted_idx, float4* out, hipStream_t stream, const unsigned* d_totals = nullptr,
                                  float4* d_big_pts = nullptr /* n points of scratch for crowded voxels */);
// order_range's radix form: see ndt_grid_kernels.hip
int order_radix_passes(long long n_cells);
size_t order_radix_cntmat_words(long long n_cells, int n, int* n_blocks = nullptr, int* ppb = nullptr, int* digit_bits = nullptr);
hipError_t launch_order_radix(const float4* pts, int n, const GridGeom& g, unsigned* cntmat, unsigned* bucket_base, float4* tmp, float4* out,
                              unsigned* counts, hipStream_t stream);
hipError_t launch_sort_gather(const float4* pts, const unsigned* leaf_start, const int* leaf_count, int n_leaves,
                              int* sorted_idx, float4* out, hipStream_t stream, const unsigned* d_totals = nullptr);
hipError_t launch_derivatives_stamped(const float4* src, int n, const GridView& gv, const EvalParams& P, int n_blocks,
                                      double* partials, unsigned long long* stamps, hipStream_t stream);
hipError_t launch_selftest_reduce(int n_blocks, double* out, hipStream_t stream);
int derivative_blocks(int n, int search);  // grid size used for n source points
int batch_blocks(int n);                   // blocks per scan inside a lock-step batch (a function of the scan's size only)
int derivative_variant();
int scan_tiles(long long n_cells);  // number of 2048-cell tiles of the cell scan

}  // namespace ndt
