// Test program for include/pclomp/voxel_grid_covariance_omp.h (compiled against tests/pcl_stub, linked with libndt_mi355.so):
// builds the grid of a cloud the way a caller of the reference's class would and prints what the queries return.
//   vgc_harness cloud.f32 n_points leaf n_queries
#include <pclomp/voxel_grid_covariance_omp.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>

typedef pcl::PointXYZ P;
typedef pclomp::VoxelGridCovariance<P> Grid;

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const size_t n = std::strtoull(argv[2], nullptr, 10);
  const float leaf = std::strtof(argv[3], nullptr);
  const size_t nq = std::strtoull(argv[4], nullptr, 10);
  std::vector<float> xyz(3 * n);
  std::ifstream(argv[1], std::ios::binary).read(reinterpret_cast<char*>(xyz.data()), static_cast<std::streamsize>(xyz.size() * sizeof(float)));
  pcl::PointCloud<P>::Ptr cloud(new pcl::PointCloud<P>);
  cloud->points.resize(n);
  for (size_t i = 0; i < n; i++) {
    cloud->points[i].x = xyz[3 * i];
    cloud->points[i].y = xyz[3 * i + 1];
    cloud->points[i].z = xyz[3 * i + 2];
    cloud->points[i].pad = 1.f;
  }
  cloud->width = static_cast<unsigned>(n);
  Grid g;
  g.setLeafSize(leaf, leaf, leaf);
  g.setInputCloud(cloud);
  g.setMinPointPerVoxel(2);  // warns and takes 3 (.h:228-239)
  std::printf("min_points_clamped %d\n", g.getMinPointPerVoxel());
  g.setMinPointPerVoxel(6);
  pcl::PointCloud<P> out;
  g.filter(out, true);
  const Grid::Map& leaves = g.getLeaves();
  std::printf("leaves %zu output %zu centroids %zu\n", leaves.size(), out.points.size(), g.getCentroids()->points.size());
  // every leaf: index, count, mean; candidates: covariance, inverse, eigenvalues, and how good the eigenbasis is
  double worst_resid = 0, worst_ortho = 0;
  for (Grid::Map::const_iterator it = leaves.begin(); it != leaves.end(); ++it) {
    const Grid::Leaf& l = it->second;
    std::printf("leaf %zu %d %.17g %.17g %.17g", it->first, l.getPointCount(), l.getMean()[0], l.getMean()[1], l.getMean()[2]);
    if (l.nr_points >= 6 || l.nr_points == -1) {
      const Eigen::Matrix3d c = l.getCov(), ic = l.getInverseCov(), v = l.getEvecs();
      const Eigen::Vector3d e = l.getEvals();
      std::printf(" %.17g %.17g %.17g %.17g %.17g %.17g | %.17g %.17g %.17g | %.17g %.17g %.17g", c(0, 0), c(0, 1), c(0, 2), c(1, 1), c(1, 2), c(2, 2),
                  ic(0, 0), ic(1, 1), ic(2, 2), e[0], e[1], e[2]);
      if (l.nr_points >= 6)
        for (int k = 0; k < 3; k++) {
          double scale = std::fabs(e[2]) + 1e-300;
          for (int r = 0; r < 3; r++) {
            double cv = 0;
            for (int q = 0; q < 3; q++) cv += c(r, q) * v(q, k);
            worst_resid = std::max(worst_resid, std::fabs(cv - e[k] * v(r, k)) / scale);
          }
          for (int k2 = 0; k2 < 3; k2++) {
            double d = 0;
            for (int r = 0; r < 3; r++) d += v(r, k) * v(r, k2);
            worst_ortho = std::max(worst_ortho, std::fabs(d - (k == k2 ? 1.0 : 0.0)));
          }
        }
    }
    std::printf("\n");
  }
  std::printf("eigenbasis %.3g %.3g\n", worst_resid, worst_ortho);
  for (size_t i = 0; i < out.points.size() && i < 5; i++) std::printf("centroid %zu %.9g %.9g %.9g\n", i, out.points[i].x, out.points[i].y, out.points[i].z);
  // queries at the first nq input points, moved a little so that some land in neighbouring cells
  for (size_t i = 0; i < nq && i < n; i++) {
    P p = cloud->points[i];
    p.x += 0.37f * static_cast<float>(static_cast<int>(i % 5) - 2);
    p.y -= 0.21f * static_cast<float>(static_cast<int>(i % 3) - 1);
    std::vector<Grid::LeafConstPtr> nb;
    const int n26 = g.getNeighborhoodAtPoint(p, nb);
    const int n7 = g.getNeighborhoodAtPoint7(p, nb);
    double m7 = 0;
    for (size_t k = 0; k < nb.size(); k++) m7 += nb[k]->getMean()[0] * static_cast<double>(k + 1);
    const int n1 = g.getNeighborhoodAtPoint1(p, nb);
    Grid::LeafConstPtr own = g.getLeaf(p);
    std::vector<float> d2;
    const int nr = g.radiusSearch(p, leaf, nb, d2);
    const float r0 = nr ? d2[0] : -1.f;
    const int nk = g.nearestKSearch(p, 3, nb, d2);
    std::printf("query %zu %d %d %.17g %d %d %d %.9g %d %.9g %.9g\n", i, n26, n7, m7, n1, own ? own->getPointCount() : -999, nr, r0, nk,
                nk > 0 ? d2[0] : -1.f, nk > 2 ? d2[2] : -1.f);
  }
  // value semantics: the copy answers from its own leaves
  Grid copy(g);
  std::vector<Grid::LeafConstPtr> nb;
  std::printf("copy %zu %d\n", copy.getLeaves().size(), copy.getNeighborhoodAtPoint7(cloud->points[0], nb));
  // getDisplayCloud (.h:407-412): 1000 draws from every valid voxel's normal distribution -- their sample mean and the spread
  // along the axes must be the voxel's own (checked for the first valid voxel: mean within 5 standard errors)
  {
    pcl::PointCloud<pcl::PointXYZ> disp;
    g.getDisplayCloud(disp);
    size_t n_valid = 0;
    const Grid::Leaf* first = nullptr;
    for (const auto& kv : g.getLeaves())
      if (kv.second.nr_points >= g.getMinPointPerVoxel()) {
        if (!first) first = &kv.second;
        n_valid++;
      }
    double m[3] = {0, 0, 0};
    for (int i = 0; i < 1000 && first; i++) { m[0] += disp.points[i].x; m[1] += disp.points[i].y; m[2] += disp.points[i].z; }
    const double scale = std::sqrt(3.0) * leaf;  // |leaf_size| multiplies the unit draws (the reference's quirk)
    double worst = 0;
    for (int k = 0; k < 3 && first; k++) worst = std::max(worst, std::fabs(m[k] / 1000 - first->mean_(k)) / (scale * std::sqrt(first->cov_(k, k) / 1000.0)));
    std::printf("display %zu %zu %.3f\n", disp.points.size(), n_valid, worst);
  }
  return 0;
}
