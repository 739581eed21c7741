// TEST-ONLY: a caller shaped like ndt_omp/apps/align.cpp:14-33 and
// lidar_subscriber/src/ndt_omp_mapping_node.cpp:151-169, compiled against include/pclomp/ndt_omp.h
// (with the PCL stand-ins of tests/pcl_stub) and linked to libndt_mi355.so.
//   adapter_harness target.f32 n_target source.f32 n_source   (xyz float32 triples)
#include <pclomp/gicp_omp.h>
#include <pclomp/ndt_omp.h>

#include <cstdio>
#include <cstdlib>

typedef pcl::PointXYZ PointT;
typedef pclomp::NormalDistributionsTransform<PointT, PointT> NDT;

static pcl::PointCloud<PointT>::Ptr load(const char* path, size_t n) {
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
  std::vector<float> raw(n * 3);
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(raw.data(), sizeof(float), raw.size(), f) != raw.size()) {
    std::fprintf(stderr, "cannot read %s\n", path);
    std::exit(2);
  }
  std::fclose(f);
  c->points.resize(n);
  for (size_t i = 0; i < n; i++) c->points[i] = PointT{raw[3 * i], raw[3 * i + 1], raw[3 * i + 2], 1.0f};
  c->width = static_cast<unsigned>(n);
  return c;
}

// ndt_omp_mapping_node.cpp:151-169: configures a member, aligns, returns the object BY VALUE
static NDT align_consecutive(NDT& ndt, const pcl::PointCloud<PointT>::Ptr& target, const pcl::PointCloud<PointT>::Ptr& source) {
  ndt.setInputTarget(target);
  ndt.setInputSource(source);
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  ndt.align(*aligned);
  return ndt;
}

static void print(const char* tag, const Eigen::Matrix4f& m, bool conv, int iters) {
  std::printf("%s converged=%d iterations=%d T=", tag, conv ? 1 : 0, iters);
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) std::printf("%.9g ", m(r, c));
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (argc != 5) return 2;
  auto target = load(argv[1], std::strtoul(argv[2], nullptr, 10));
  auto source = load(argv[3], std::strtoul(argv[4], nullptr, 10));

  // apps/align.cpp:95-103 -- through the pcl::Registration base pointer
  NDT::Ptr ndt_omp(new NDT());
  ndt_omp->setResolution(1.0);
  ndt_omp->setNumThreads(8);
  ndt_omp->setNeighborhoodSearchMethod(pclomp::DIRECT7);
  pcl::Registration<PointT, PointT>::Ptr registration = ndt_omp;
  registration->setInputTarget(target);
  registration->setInputSource(source);
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>());
  registration->align(*aligned);
  print("align_app", registration->getFinalTransformation(), registration->hasConverged(), ndt_omp->getFinalNumIteration());
  std::printf("aligned0 %.9g %.9g %.9g %.9g\n", aligned->points[0].x, aligned->points[0].y, aligned->points[0].z, aligned->points[0].pad);

  // ndt_omp_mapping_node.cpp:55-62 parameters, object returned by value
  NDT member;
  member.setResolution(1.0);
  member.setStepSize(0.1);
  member.setTransformationEpsilon(0.01);
  member.setMaximumIterations(64);
  member.setNumThreads(40);
  member.setNeighborhoodSearchMethod(pclomp::DIRECT7);
  NDT copy = align_consecutive(member, target, source);
  print("mapping_node", copy.getFinalTransformation(), copy.hasConverged(), copy.getFinalNumIteration());

  // ndt_rosbag_mapping_node.cpp:124-141: previous result as the initial guess
  pcl::PointCloud<PointT>::Ptr aligned2(new pcl::PointCloud<PointT>());
  copy.align(*aligned2, copy.getFinalTransformation());
  print("rosbag_node", copy.getFinalTransformation(), copy.hasConverged(), copy.getFinalNumIteration());
  std::printf("trans_probability %.12g\n", copy.getTransformationProbability());
  // ndt_rosbag_mapping_node.cpp:133 prints the fitness of the derived object
  std::printf("fitness %.12g\n", copy.getFitnessScore());

  // A caller that refills its cloud object in place and hands the SAME pointer over again (legal PCL usage; the
  // reference reads *input_ at every align, ndt_omp_impl.hpp:833): the new points must be registered, not the old upload.
  {
    pcl::PointCloud<PointT>::Ptr reused(new pcl::PointCloud<PointT>(*source));
    NDT refill;
    refill.setResolution(1.0);
    refill.setInputTarget(target);
    refill.setInputSource(reused);
    pcl::PointCloud<PointT>::Ptr a(new pcl::PointCloud<PointT>());
    refill.align(*a);
    for (auto& p : reused->points) {  // same object, other points: shifted by (0.4, -0.3, 0.05)
      p.x += 0.4f;
      p.y -= 0.3f;
      p.z += 0.05f;
    }
    refill.setInputSource(reused);
    refill.align(*a);
    print("refill_same_ptr", refill.getFinalTransformation(), refill.hasConverged(), refill.getFinalNumIteration());
    refill.align(*a);  // no set call in between: the upload is reused, the answer is the same
    print("refill_again", refill.getFinalTransformation(), refill.hasConverged(), refill.getFinalNumIteration());
  }

  // apps/align.cpp:84-86 -- pclomp::GICP through the pcl::Registration base pointer
  typedef pclomp::GeneralizedIterativeClosestPoint<PointT, PointT> GICP;
  GICP::Ptr gicp_omp(new GICP());
  pcl::Registration<PointT, PointT>::Ptr reg2 = gicp_omp;
  reg2->setInputTarget(target);
  reg2->setInputSource(source);
  pcl::PointCloud<PointT>::Ptr aligned3(new pcl::PointCloud<PointT>());
  reg2->align(*aligned3);
  print("gicp_app", reg2->getFinalTransformation(), reg2->hasConverged(), 0);
  std::printf("gicp_aligned0 %.9g %.9g %.9g %.9g\n", aligned3->points[0].x, aligned3->points[0].y, aligned3->points[0].z, aligned3->points[0].pad);
  std::printf("gicp_fitness %.12g\n", gicp_omp->getFitnessScore());
  // gicp_omp.h:165-168,186-189 -- caller-supplied covariances (isotropic 0.01 I on both clouds), used until the clouds are set again
  {
    GICP::MatricesVectorPtr cs(new GICP::MatricesVector(source->points.size())), ct(new GICP::MatricesVector(target->points.size()));
    for (auto* v : {cs.get(), ct.get()})
      for (auto& m : *v) {
        for (int i = 0; i < 9; i++) m(i) = 0.0;
        m(0, 0) = m(1, 1) = m(2, 2) = 0.01;
      }
    gicp_omp->setSourceCovariances(cs);
    gicp_omp->setTargetCovariances(ct);
    pcl::PointCloud<PointT>::Ptr aligned4(new pcl::PointCloud<PointT>());
    reg2->align(*aligned4);
    print("gicp_user_cov", reg2->getFinalTransformation(), reg2->hasConverged(), 0);
    reg2->setInputSource(source);  // resets the source's covariances: k-NN ones again, the target keeps the supplied ones
    reg2->align(*aligned4);
    print("gicp_mixed_cov", reg2->getFinalTransformation(), reg2->hasConverged(), 0);
    // gicp_omp_impl.hpp:386-397: an empty pointer means "compute them": a null setter call after an earlier supplied set
    // must bring the k-NN covariances back (the first GICP registration above), not keep the stale matrices
    gicp_omp->setTargetCovariances(GICP::MatricesVectorPtr());
    reg2->align(*aligned4);
    print("gicp_cleared_cov", reg2->getFinalTransformation(), reg2->hasConverged(), 0);
  }
  return 0;
}
