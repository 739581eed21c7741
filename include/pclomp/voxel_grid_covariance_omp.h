// pclomp/voxel_grid_covariance_omp.h -- header-only adapter: pclomp::VoxelGridCovariance<PointT>, the public type of
// the reference's target grid, re-declared on top of the MI355X C-ABI (include/ndt_mi355.h, libndt_mi355.so).
//
// Replaces, with the same include path, class name, base class, Leaf struct and method surface:
//   /root/reference/ndt_omp/include/pclomp/voxel_grid_covariance_omp.h:59-556 (+ voxel_grid_covariance_omp_impl.hpp,
//   src/pclomp/voxel_grid_covariance_omp.cpp, which this header makes unnecessary).
// No ToySLAM node includes it (the nodes reach the grid through NormalDistributionsTransform, whose adapter keeps the grid
// on the device); it is here for code that inspects the voxels: filter() builds the grid on the GPU (K1, the kernels
// setInputTarget runs), ndt_grid_dump brings the leaves back in the order of the reference's std::map, and the queries
// (getLeaf, getNeighborhoodAtPoint*, radiusSearch, nearestKSearch) are answered from that host copy with the reference's
// index arithmetic (_impl.hpp:372-444, .h:309-375).
//
// Differences a caller can observe:
//   * cubic leaves only (setLeafSize(l, l, l)): the C-ABI's grid has one resolution, as every NDT caller's does;
//   * only x, y, z are averaged (downsample_all_data_ = false, the constructor's own setting, .h:216): Leaf::centroid
//     has three entries;
//   * Leaf::evecs_ is an orthonormal eigenbasis of the leaf's covariance (cyclic Jacobi on the host, columns in the order
//     of evals_): the device keeps eigenvalues only, and eigenvectors are defined up to sign (and up to a rotation inside an
//     eigenspace the inflation of :345-356 made degenerate), so they are not bit-comparable with Eigen's in any case;
//   * radiusSearch / nearestKSearch scan the centroids (exact, ascending distance like the kd-tree's sorted results; ties in
//     leaf order) -- the evaluation kernels have their own search, this one serves inspection;
//   * a leaf with fewer than min_points_per_voxel_ points carries its count and mean; its cov_ / icov_ / evecs_ / evals_ stay
//     at the constructor's values (the reference leaves its running sums there);
//   * getDisplayCloud is not provided (boost::random sampling for a viewer; SURVEY section 2: out of scope).
#ifndef PCL_VOXEL_GRID_COVARIANCE_OMP_MI355_H_
#define PCL_VOXEL_GRID_COVARIANCE_OMP_MI355_H_

#include <pcl/filters/voxel_grid.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ndt_mi355.h"

namespace pclomp {

template <typename PointT>
class VoxelGridCovariance : public pcl::VoxelGrid<PointT> {
 protected:
  using pcl::VoxelGrid<PointT>::filter_name_;
  using pcl::VoxelGrid<PointT>::getClassName;
  using pcl::VoxelGrid<PointT>::input_;
  using pcl::VoxelGrid<PointT>::downsample_all_data_;
  using pcl::VoxelGrid<PointT>::save_leaf_layout_;
  using pcl::VoxelGrid<PointT>::leaf_size_;
  using pcl::VoxelGrid<PointT>::inverse_leaf_size_;
  using pcl::VoxelGrid<PointT>::min_b_;
  using pcl::VoxelGrid<PointT>::max_b_;
  using pcl::VoxelGrid<PointT>::div_b_;
  using pcl::VoxelGrid<PointT>::divb_mul_;

  typedef pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::Ptr PointCloudPtr;
  typedef typename PointCloud::ConstPtr PointCloudConstPtr;

 public:
#if defined(PCL_VERSION_CALC)
#if PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
#define NDT_MI355_VGC_PCL_SHARED_PTR 1
#endif
#endif
#ifdef NDT_MI355_VGC_PCL_SHARED_PTR
  typedef pcl::shared_ptr<pcl::VoxelGrid<PointT> > Ptr;
  typedef pcl::shared_ptr<const pcl::VoxelGrid<PointT> > ConstPtr;
#else
  typedef boost::shared_ptr<pcl::VoxelGrid<PointT> > Ptr;
  typedef boost::shared_ptr<const pcl::VoxelGrid<PointT> > ConstPtr;
#endif

  /** voxel_grid_covariance_omp.h:98-190, member for member. */
  struct Leaf {
    Leaf() : nr_points(0) {
      mean_.setZero();
      cov_ = Eigen::Matrix3d::Identity();
      icov_.setZero();
      evecs_ = Eigen::Matrix3d::Identity();
      evals_.setZero();
    }
    Eigen::Matrix3d getCov() const { return cov_; }
    Eigen::Matrix3d getInverseCov() const { return icov_; }
    Eigen::Vector3d getMean() const { return mean_; }
    Eigen::Matrix3d getEvecs() const { return evecs_; }
    Eigen::Vector3d getEvals() const { return evals_; }
    int getPointCount() const { return nr_points; }
    int nr_points;
    Eigen::Vector3d mean_;
    Eigen::VectorXf centroid;
    Eigen::Matrix3d cov_;
    Eigen::Matrix3d icov_;
    Eigen::Matrix3d evecs_;
    Eigen::Vector3d evals_;
  };
  typedef Leaf* LeafPtr;
  typedef const Leaf* LeafConstPtr;
  typedef std::map<size_t, Leaf> Map;

  /** .h:208-225 */
  VoxelGridCovariance() : searchable_(true), min_points_per_voxel_(6), min_covar_eigvalue_mult_(0.01), handle_(nullptr) {
    downsample_all_data_ = false;
    save_leaf_layout_ = false;
    leaf_size_.setZero();
    min_b_.setZero();
    max_b_.setZero();
    filter_name_ = "VoxelGridCovariance";
    voxel_centroids_.reset(new PointCloud);
  }
  VoxelGridCovariance(const VoxelGridCovariance& o) : pcl::VoxelGrid<PointT>(o), handle_(nullptr) { copy_from(o); }
  VoxelGridCovariance& operator=(const VoxelGridCovariance& o) {
    if (this != &o) {
      pcl::VoxelGrid<PointT>::operator=(o);
      copy_from(o);
    }
    return *this;
  }
  ~VoxelGridCovariance() {
    if (handle_) ndt_destroy(handle_);
  }

  /** .h:228-239 */
  inline void setMinPointPerVoxel(int min_points_per_voxel) {
    if (min_points_per_voxel > 2) {
      min_points_per_voxel_ = min_points_per_voxel;
    } else {
      PCL_WARN("%s: Covariance calculation requires at least 3 points, setting Min Point per Voxel to 3 ", this->getClassName().c_str());
      min_points_per_voxel_ = 3;
    }
  }
  inline int getMinPointPerVoxel() { return min_points_per_voxel_; }
  /** .h:253-266 */
  inline void setCovEigValueInflationRatio(double min_covar_eigvalue_mult) { min_covar_eigvalue_mult_ = min_covar_eigvalue_mult; }
  inline double getCovEigValueInflationRatio() { return min_covar_eigvalue_mult_; }

  /** .h:273-303: output = the centroids of the voxels with at least min_points_per_voxel_ points, in leaf order. */
  inline void filter(PointCloud& output, bool searchable = false) {
    searchable_ = searchable;
    applyFilter(output);
    voxel_centroids_ = PointCloudPtr(new PointCloud(output));
  }
  inline void filter(bool searchable = false) {
    searchable_ = searchable;
    voxel_centroids_ = PointCloudPtr(new PointCloud);
    applyFilter(*voxel_centroids_);
  }

  /** .h:309-375 */
  inline LeafConstPtr getLeaf(int index) { return find_leaf(static_cast<long long>(index)); }
  inline LeafConstPtr getLeaf(PointT& p) { return leaf_at(p.x, p.y, p.z); }
  inline LeafConstPtr getLeaf(Eigen::Vector3f& p) { return leaf_at(p[0], p[1], p[2]); }

  /** _impl.hpp:372-444.  relative_coordinates: 3 x n integer offsets, as (dx, dy, dz) triples. */
  int getNeighborhoodAtPoint(const std::vector<int>& relative_coordinates_xyz, const PointT& reference_point,
                             std::vector<LeafConstPtr>& neighbors) const {
    neighbors.clear();
    // floor(x / leaf) here, floor(x * inverse_leaf) while building: the reference's own asymmetry (_impl.hpp:379-381 vs :218-223)
    const int ijk[3] = {static_cast<int>(std::floor(reference_point.x / leaf_size_[0])), static_cast<int>(std::floor(reference_point.y / leaf_size_[1])),
                        static_cast<int>(std::floor(reference_point.z / leaf_size_[2]))};
    const size_t n = relative_coordinates_xyz.size() / 3;
    neighbors.reserve(n);
    for (size_t ni = 0; ni < n; ni++) {
      bool inside = true;
      long long idx = 0;
      for (int a = 0; a < 3; a++) {
        const int d = relative_coordinates_xyz[3 * ni + a];
        if (min_b_[a] - ijk[a] > d || max_b_[a] - ijk[a] < d) inside = false;
        idx += static_cast<long long>(ijk[a] + d - min_b_[a]) * divb_mul_[a];
      }
      if (!inside) continue;
      typename Map::const_iterator it = leaves_.find(static_cast<size_t>(idx));
      if (it != leaves_.end() && it->second.nr_points >= min_points_per_voxel_) neighbors.push_back(&(it->second));
    }
    return static_cast<int>(neighbors.size());
  }
  /** the reference's signature (Eigen::MatrixXi, 3 x n: one offset per column) */
  template <class MatrixXi>
  int getNeighborhoodAtPoint(const MatrixXi& relative_coordinates, const PointT& reference_point, std::vector<LeafConstPtr>& neighbors) const {
    std::vector<int> rel;
    rel.reserve(3 * static_cast<size_t>(relative_coordinates.cols()));
    for (int ni = 0; ni < static_cast<int>(relative_coordinates.cols()); ni++)
      for (int a = 0; a < 3; a++) rel.push_back(relative_coordinates(a, ni));
    return getNeighborhoodAtPoint(rel, reference_point, neighbors);
  }
  /** the 26 cells around the point's own ([PCL] getAllNeighborCellIndices: 13 half offsets, then their negatives) */
  int getNeighborhoodAtPoint(const PointT& reference_point, std::vector<LeafConstPtr>& neighbors) const {
    static const int half[13][3] = {{-1, -1, -1}, {-1, 0, -1}, {-1, 1, -1}, {0, -1, -1}, {0, 0, -1}, {0, 1, -1}, {1, -1, -1},
                                    {1, 0, -1},   {1, 1, -1},  {-1, -1, 0}, {0, -1, 0},  {1, -1, 0}, {-1, 0, 0}};
    std::vector<int> rel;
    rel.reserve(78);
    for (int k = 0; k < 13; k++) rel.insert(rel.end(), half[k], half[k] + 3);
    for (int k = 0; k < 13; k++)
      for (int a = 0; a < 3; a++) rel.push_back(-half[k][a]);
    return getNeighborhoodAtPoint(rel, reference_point, neighbors);
  }
  int getNeighborhoodAtPoint7(const PointT& reference_point, std::vector<LeafConstPtr>& neighbors) const {
    static const int rel7[21] = {0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1};
    return getNeighborhoodAtPoint(std::vector<int>(rel7, rel7 + 21), reference_point, neighbors);
  }
  int getNeighborhoodAtPoint1(const PointT& reference_point, std::vector<LeafConstPtr>& neighbors) const {
    return getNeighborhoodAtPoint(std::vector<int>(3, 0), reference_point, neighbors);
  }

  /** .h:391-405 */
  inline const Map& getLeaves() { return leaves_; }
  inline PointCloudPtr getCentroids() { return voxel_centroids_; }

  /** .h:407-412, _impl.hpp:449-490: a cloud for display -- 1000 points per voxel with enough points, drawn from the voxel's
   *  normal distribution (mean + L r, L the Cholesky factor of the covariance, r three draws from N(0, |leaf_size|)).  Host
   *  code over the dumped leaves.  The reference draws with boost::mt19937 + boost::normal_distribution; here std::mt19937
   *  (the same engine and seed) + std::normal_distribution: the same distribution, not the same sample. */
  void getDisplayCloud(pcl::PointCloud<pcl::PointXYZ>& cell_cloud) {
    cell_cloud.points.clear();
    const int pnt_per_cell = 1000;
    std::mt19937 rng;
    const double lx = leaf_size_[0], ly = leaf_size_[1], lz = leaf_size_[2];
    std::normal_distribution<double> nd(0.0, std::sqrt(lx * lx + ly * ly + lz * lz));
    for (typename Map::const_iterator it = leaves_.begin(); it != leaves_.end(); ++it) {
      const Leaf& leaf = it->second;
      if (leaf.nr_points < min_points_per_voxel_) continue;
      // lower Cholesky factor of the (inflated) covariance, as Eigen::LLT computes it column by column
      double L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int j = 0; j < 3; j++) {
        double d = leaf.cov_(j, j);
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
        L[j][j] = std::sqrt(d);
        for (int i = j + 1; i < 3; i++) {
          double v = leaf.cov_(i, j);
          for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
          L[i][j] = v / L[j][j];
        }
      }
      for (int i = 0; i < pnt_per_cell; i++) {
        const double r0 = nd(rng), r1 = nd(rng), r2 = nd(rng);
        pcl::PointXYZ p;
        p.x = static_cast<float>(leaf.mean_(0) + L[0][0] * r0);
        p.y = static_cast<float>(leaf.mean_(1) + L[1][0] * r0 + L[1][1] * r1);
        p.z = static_cast<float>(leaf.mean_(2) + L[2][0] * r0 + L[2][1] * r1 + L[2][2] * r2);
        cell_cloud.points.push_back(p);
      }
    }
    cell_cloud.width = static_cast<unsigned>(cell_cloud.points.size());
    cell_cloud.height = 1;
  }

  /** .h:423-465: the k voxels (of those in the centroid cloud) whose centroids are nearest to the point, ascending. */
  int nearestKSearch(const PointT& point, int k, std::vector<LeafConstPtr>& k_leaves, std::vector<float>& k_sqr_distances) {
    k_leaves.clear();
    k_sqr_distances.clear();
    if (!searchable_) {
      PCL_WARN("%s: Not Searchable", this->getClassName().c_str());
      return 0;
    }
    std::vector<std::pair<float, int> > d;
    centroid_distances(point, d);
    const size_t kk = std::min(d.size(), static_cast<size_t>(std::max(k, 0)));
    std::partial_sort(d.begin(), d.begin() + kk, d.end());
    for (size_t i = 0; i < kk; i++) {
      k_leaves.push_back(&leaves_[static_cast<size_t>(voxel_centroids_leaf_indices_[d[i].second])]);
      k_sqr_distances.push_back(d[i].first);
    }
    return static_cast<int>(kk);
  }
  inline int nearestKSearch(const PointCloud& cloud, int index, int k, std::vector<LeafConstPtr>& k_leaves, std::vector<float>& k_sqr_distances) {
    if (index >= static_cast<int>(cloud.points.size()) || index < 0) return 0;
    return nearestKSearch(cloud.points[index], k, k_leaves, k_sqr_distances);
  }
  /** .h:476-525: every voxel of the centroid cloud whose centroid is closer than radius (dist < r^2), ascending. */
  int radiusSearch(const PointT& point, double radius, std::vector<LeafConstPtr>& k_leaves, std::vector<float>& k_sqr_distances,
                   unsigned int max_nn = 0) const {
    k_leaves.clear();
    k_sqr_distances.clear();
    if (!searchable_) {
      PCL_WARN("%s: Not Searchable", this->getClassName().c_str());
      return 0;
    }
    std::vector<std::pair<float, int> > d;
    centroid_distances(point, d);
    const float r2 = static_cast<float>(radius * radius);
    std::vector<std::pair<float, int> > hit;
    for (size_t i = 0; i < d.size(); i++)
      if (d[i].first < r2) hit.push_back(d[i]);
    std::sort(hit.begin(), hit.end());
    if (max_nn > 0 && hit.size() > max_nn) hit.resize(max_nn);
    for (size_t i = 0; i < hit.size(); i++) {
      typename Map::const_iterator it = leaves_.find(static_cast<size_t>(voxel_centroids_leaf_indices_[hit[i].second]));
      k_leaves.push_back(&(it->second));
      k_sqr_distances.push_back(hit[i].first);
    }
    return static_cast<int>(hit.size());
  }
  inline int radiusSearch(const PointCloud& cloud, int index, double radius, std::vector<LeafConstPtr>& k_leaves,
                          std::vector<float>& k_sqr_distances, unsigned int max_nn = 0) const {
    if (index >= static_cast<int>(cloud.points.size()) || index < 0) return 0;
    return radiusSearch(cloud.points[index], radius, k_leaves, k_sqr_distances, max_nn);
  }

 protected:
  /** _impl.hpp:48-370 on the device; the leaves come back through ndt_grid_dump. */
  void applyFilter(PointCloud& output) {
    voxel_centroids_leaf_indices_.clear();
    leaves_.clear();
    output.points.clear();
    output.height = 1;
    output.is_dense = true;
    if (!input_) {
      PCL_WARN("[%s::applyFilter] No input dataset given!\n", getClassName().c_str());
      output.width = output.height = 0;
      return;
    }
    if (leaf_size_[0] != leaf_size_[1] || leaf_size_[0] != leaf_size_[2] || !(leaf_size_[0] > 0.f))
      throw std::runtime_error("pclomp::VoxelGridCovariance (MI355X): cubic leaves only -- setLeafSize(l, l, l)");
    if (!handle_) check(ndt_create(default_device(), &handle_), "ndt_create");
    check(ndt_set_resolution(handle_, leaf_size_[0]), "ndt_set_resolution");
    check(ndt_set_min_points_per_voxel(handle_, min_points_per_voxel_), "ndt_set_min_points_per_voxel");
    check(ndt_set_cov_eig_value_inflation_ratio(handle_, min_covar_eigvalue_mult_), "ndt_set_cov_eig_value_inflation_ratio");
    check(ndt_set_input_target(handle_, input_->points.data(), input_->points.size(), sizeof(PointT), input_->is_dense ? 1 : 0),
          "ndt_set_input_target");
    size_t n_leaves = 0, n_valid = 0;
    check(ndt_grid_size(handle_, &n_leaves, &n_valid), "ndt_grid_size");
    int mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
    check(ndt_grid_info(handle_, mn, mx, dv), "ndt_grid_info");
    for (int a = 0; a < 3; a++) {
      min_b_[a] = mn[a];
      max_b_[a] = mx[a];
      div_b_[a] = dv[a];
    }
    min_b_[3] = max_b_[3] = div_b_[3] = 0;
    divb_mul_[0] = 1;  // :103
    divb_mul_[1] = div_b_[0];
    divb_mul_[2] = div_b_[0] * div_b_[1];
    divb_mul_[3] = 0;
    output.width = 0;
    if (n_leaves == 0) return;
    std::vector<int64_t> idx(n_leaves);
    std::vector<int> cnt(n_leaves);
    std::vector<double> mean(3 * n_leaves), cov(9 * n_leaves), icov(9 * n_leaves), evals(3 * n_leaves);
    check(ndt_grid_dump(handle_, idx.data(), cnt.data(), mean.data(), cov.data(), icov.data(), evals.data()), "ndt_grid_dump");
    output.points.reserve(n_leaves);
    typename Map::iterator hint = leaves_.end();
    for (size_t i = 0; i < n_leaves; i++) {  // ascending index: the order of the reference's std::map
      hint = leaves_.insert(hint, std::make_pair(static_cast<size_t>(idx[i]), Leaf()));
      Leaf& leaf = hint->second;
      leaf.nr_points = cnt[i];
      leaf.centroid.resize(3);
      for (int a = 0; a < 3; a++) {
        leaf.mean_[a] = mean[3 * i + a];
        leaf.centroid[a] = static_cast<float>(mean[3 * i + a]);
      }
      // a voxel that reached min_points_per_voxel_ is in the output and the centroid cloud even when it was rejected
      // afterwards (nr_points = -1, :337-341,360-364): the reference pushes the centroid before it looks at the eigenvalues
      const bool candidate = cnt[i] >= min_points_per_voxel_ || cnt[i] == -1;
      if (!candidate) continue;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
          leaf.cov_(r, c) = cov[9 * i + 3 * r + c];
          leaf.icov_(r, c) = icov[9 * i + 3 * r + c];
        }
      for (int a = 0; a < 3; a++) leaf.evals_[a] = evals[3 * i + a];
      eigenbasis(leaf.cov_, leaf.evals_, leaf.evecs_);
      PointT p = PointT();
      p.x = leaf.centroid[0];
      p.y = leaf.centroid[1];
      p.z = leaf.centroid[2];
      output.points.push_back(p);
      if (searchable_) voxel_centroids_leaf_indices_.push_back(static_cast<int>(idx[i]));
    }
    output.width = static_cast<unsigned>(output.points.size());
  }

  bool searchable_;
  int min_points_per_voxel_;
  double min_covar_eigvalue_mult_;
  Map leaves_;
  PointCloudPtr voxel_centroids_;
  std::vector<int> voxel_centroids_leaf_indices_;

 private:
  ndt_handle handle_;

  static int default_device() {
    const char* v = std::getenv("NDT_MI355_DEVICE");
    return v ? std::atoi(v) : 0;
  }
  static void check(ndt_status s, const char* what) {
    if (s != NDT_OK) throw std::runtime_error(std::string("pclomp::VoxelGridCovariance (MI355X): ") + what + ": " + ndt_last_error());
  }
  void copy_from(const VoxelGridCovariance& o) {  // the host copy of the leaves is the state; a device handle is made on demand
    searchable_ = o.searchable_;
    min_points_per_voxel_ = o.min_points_per_voxel_;
    min_covar_eigvalue_mult_ = o.min_covar_eigvalue_mult_;
    leaves_ = o.leaves_;
    voxel_centroids_ = o.voxel_centroids_;
    voxel_centroids_leaf_indices_ = o.voxel_centroids_leaf_indices_;
  }
  LeafConstPtr find_leaf(long long idx) const {
    typename Map::const_iterator it = leaves_.find(static_cast<size_t>(idx));
    return it != leaves_.end() ? &(it->second) : nullptr;
  }
  // .h:326-346: floor(x * inverse_leaf) - min_b, no bounds test (a point outside the box simply finds no leaf, or another's)
  LeafConstPtr leaf_at(float x, float y, float z) const {
    const int i0 = static_cast<int>(std::floor(x * inverse_leaf_size_[0]) - static_cast<float>(min_b_[0]));
    const int i1 = static_cast<int>(std::floor(y * inverse_leaf_size_[1]) - static_cast<float>(min_b_[1]));
    const int i2 = static_cast<int>(std::floor(z * inverse_leaf_size_[2]) - static_cast<float>(min_b_[2]));
    return find_leaf(static_cast<long long>(i0) * divb_mul_[0] + static_cast<long long>(i1) * divb_mul_[1] + static_cast<long long>(i2) * divb_mul_[2]);
  }
  // squared f32 distances ([FLANN] L2_Simple: (dx*dx + dy*dy) + dz*dz) from the point to every centroid of the centroid cloud
  void centroid_distances(const PointT& point, std::vector<std::pair<float, int> >& d) const {
    d.clear();
    if (!voxel_centroids_) return;
    const size_t n = std::min(voxel_centroids_->points.size(), voxel_centroids_leaf_indices_.size());
    d.reserve(n);
    for (size_t i = 0; i < n; i++) {
      const PointT& c = voxel_centroids_->points[i];
      const float dx = point.x - c.x, dy = point.y - c.y, dz = point.z - c.z;
      d.push_back(std::make_pair((dx * dx + dy * dy) + dz * dz, static_cast<int>(i)));
    }
  }
  // orthonormal eigenvectors of the symmetric 3x3 `cov` (cyclic Jacobi), columns matched to `evals` (ascending, as Eigen's
  // SelfAdjointEigenSolver orders them)
  static void eigenbasis(const Eigen::Matrix3d& cov, const Eigen::Vector3d& evals, Eigen::Matrix3d& evecs) {
    double A[3][3], V[3][3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        A[r][c] = 0.5 * (cov(r, c) + cov(c, r));
        V[r][c] = r == c ? 1.0 : 0.0;
      }
    for (int sweep = 0; sweep < 60; sweep++) {
      const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
      const double dia = std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]);
      if (off <= 1e-300 || off <= dia * 1e-17) break;
      for (int pq = 0; pq < 3; pq++) {
        const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
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
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int a, int b) { return A[a][a] < A[b][b]; });
    (void)evals;
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) evecs(r, c) = V[r][order[c]];
  }
};

}  // namespace pclomp

#endif  // PCL_VOXEL_GRID_COVARIANCE_OMP_MI355_H_
