"""The header-only pclomp/ndt_omp.h adapter: compiles against test-only PCL/Eigen stand-ins (CPU),
and a caller shaped like apps/align.cpp + ndt_omp_mapping_node.cpp runs through it on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rot_err, trans_err

INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "pcl_stub")]
SRC = os.path.join(ROOT, "tests", "adapter_harness.cpp")


def test_gicp_adapter_declares_reference_surface():
    """public surface of pclomp::GeneralizedIterativeClosestPoint (gicp_omp.h:52-260) that its caller
    (apps/align.cpp:84-86) and PCL-style user code touch."""
    hdr = open(os.path.join(ROOT, "include", "pclomp", "gicp_omp.h")).read()
    for name in ["class GeneralizedIterativeClosestPoint : public pcl::IterativeClosestPoint<PointSource, PointTarget>",
                 "setInputSource", "setInputTarget", "setRotationEpsilon", "getRotationEpsilon", "setCorrespondenceRandomness",
                 "getCorrespondenceRandomness", "setMaximumOptimizerIterations", "getMaximumOptimizerIterations",
                 "computeTransformation", "max_iterations_ = 200", "transformation_epsilon_ = 5e-4", "corr_dist_threshold_ = 5."]:
        assert name in hdr, name


def test_adapter_header_compiles():
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror"] + INC + [SRC])


def test_adapter_declares_reference_surface():
    """Every public method the reference class declares (ndt_omp.h:110-238) and its callers use."""
    hdr = open(os.path.join(ROOT, "include", "pclomp", "ndt_omp.h")).read()
    for name in ["setNumThreads", "setInputTarget", "setResolution", "getResolution", "getStepSize", "setStepSize",
                 "getOutlierRatio", "setOutlierRatio", "setNeighborhoodSearchMethod", "getTransformationProbability",
                 "getFinalNumIteration", "convertTransform", "calculateScore", "computeTransformation", "search_method",
                 "EIGEN_MAKE_ALIGNED_OPERATOR_NEW", "enum NeighborSearchMethod { KDTREE, DIRECT26, DIRECT7, DIRECT1 }",
                 "public pcl::Registration<PointSource, PointTarget>"]:
        assert name in hdr, name


@pytest.mark.gpu
def test_adapter_harness_matches_oracle(built_lib, pair, tmp_path):
    from oracle import pyoracle as po
    t, s = pair
    exe = str(tmp_path / "adapter_harness")
    libdir = os.path.join(ROOT, "toyslam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + INC + [SRC, "-o", exe, "-L" + libdir, "-lndt_mi355",
                                                               "-Wl,-rpath," + libdir])
    tf, sf = str(tmp_path / "t.f32"), str(tmp_path / "s.f32")
    t.astype("<f4").tofile(tf)
    s.astype("<f4").tofile(sf)
    out = subprocess.check_output([exe, tf, str(len(t)), sf, str(len(s))], text=True)
    rows = {}
    for line in out.splitlines():
        tag, _, rest = line.partition(" ")
        rows[tag] = rest
    def parse(tag):
        f = rows[tag].split()
        conv = int(f[0].split("=")[1])
        iters = int(f[1].split("=")[1])
        T = np.array([float(x) for x in [f[2].split("=")[1]] + f[3:18]]).reshape(4, 4)
        return conv, iters, T

    o = po.OracleNDT(resolution=1.0, num_threads=8)
    o.set_target(t)
    o.set_source(s)
    r = o.align()
    conv, iters, T = parse("align_app")
    assert conv == 1 and iters == r["iterations"]
    assert rot_err(T, r["T"]) < 1e-4 and trans_err(T, r["T"]) < 1e-3
    a0 = np.array([float(x) for x in rows["aligned0"].split()])
    assert a0[3] == 1.0 and np.allclose(a0[:3], po.transform_cloud(np.c_[s[:1], [[1.0]]].astype(np.float32), r["T"])[0, :3], atol=1e-3)

    o.set(trans_eps=0.01, max_iter=64)
    r2 = o.align()
    conv, iters, T2 = parse("mapping_node")   # the by-value COPY carries the result
    assert conv == 1 and iters == r2["iterations"]
    assert rot_err(T2, r2["T"]) < 1e-4 and trans_err(T2, r2["T"]) < 1e-3
    r3 = o.align(r2["T"])
    conv, iters, T3 = parse("rosbag_node")    # previous result as the guess
    assert conv == 1 and iters == r3["iterations"]
    assert rot_err(T3, r3["T"]) < 1e-4 and trans_err(T3, r3["T"]) < 1e-3
    assert float(rows["trans_probability"]) == pytest.approx(r3["trans_probability"], rel=1e-5)
    # getFitnessScore() of the derived type runs on the GPU: mean squared nearest-neighbour distance
    from scipy.spatial import cKDTree
    moved = po.transform_cloud(np.c_[s, np.ones(len(s), np.float32)].astype(np.float32), T3.astype(np.float32))[:, :3]
    d, _ = cKDTree(t.astype(np.float64)).query(moved.astype(np.float64))
    assert float(rows["fitness"]) == pytest.approx(float(np.mean(d ** 2)), rel=1e-5)

    # setInputSource(same pointer) after the cloud was refilled in place: the NEW points are registered
    o2 = po.OracleNDT(resolution=1.0, num_threads=8)
    o2.set_target(t)
    s_shift = (s.astype(np.float32) + np.array([0.4, -0.3, 0.05], np.float32)).astype(np.float32)
    o2.set_source(s_shift)
    rr = o2.align()
    conv, iters, Tr = parse("refill_same_ptr")
    assert conv == int(rr["converged"]) and iters == rr["iterations"]
    assert rot_err(Tr, rr["T"]) < 1e-4 and trans_err(Tr, rr["T"]) < 1e-3
    assert trans_err(Tr, r["T"]) > 0.1   # (not the stale upload's answer)
    assert np.array_equal(parse("refill_again")[2], Tr)

    # pclomp::GICP through the pcl::Registration pointer (apps/align.cpp:84-86)
    og = po.OracleGICP()
    og.setInputTarget(t)
    og.setInputSource(s)
    rg = og.align(want_cloud=True)
    conv, _, Tg = parse("gicp_app")
    assert conv == int(rg["converged"]) == 1
    assert rot_err(Tg, rg["T"]) < 1e-4 and trans_err(Tg, rg["T"]) < 1e-3
    g0 = np.array([float(x) for x in rows["gicp_aligned0"].split()])
    assert g0[3] == 1.0 and np.allclose(g0[:3], rg["cloud"][0, :3], atol=1e-3)
    moved = po.transform_cloud(np.c_[s, np.ones(len(s), np.float32)].astype(np.float32), Tg.astype(np.float32))[:, :3]
    d, _ = cKDTree(t.astype(np.float64)).query(moved.astype(np.float64))
    assert float(rows["gicp_fitness"]) == pytest.approx(float(np.mean(d ** 2)), rel=1e-4)
    # setSourceCovariances / setTargetCovariances (gicp_omp.h:165-168,186-189): isotropic covariances on both clouds, then the
    # source set again (its covariances come from its neighbours again, the target keeps the supplied ones)
    iso_s, iso_t = np.tile(0.01 * np.eye(3), (len(s), 1, 1)), np.tile(0.01 * np.eye(3), (len(t), 1, 1))
    og.setSourceCovariances(iso_s)
    og.setTargetCovariances(iso_t)
    ru = og.align()
    conv, _, Tu = parse("gicp_user_cov")
    assert conv == int(ru["converged"])
    assert rot_err(Tu, ru["T"]) < 1e-4 and trans_err(Tu, ru["T"]) < 1e-3
    assert trans_err(Tu, Tg) > 1e-4  # (a different objective: it must not end exactly where the k-NN covariances end)
    og.setInputSource(s)
    rm = og.align()
    conv, _, Tm = parse("gicp_mixed_cov")
    assert conv == int(rm["converged"])
    assert rot_err(Tm, rm["T"]) < 1e-4 and trans_err(Tm, rm["T"]) < 1e-3
    # setTargetCovariances(null) after a supplied set: both clouds on their own k-NN covariances again = the first registration
    conv, _, Tc = parse("gicp_cleared_cov")
    assert conv == 1 and rot_err(Tc, rg["T"]) < 1e-4 and trans_err(Tc, rg["T"]) < 1e-3
    assert np.array_equal(Tc, Tg)


@pytest.mark.gpu
def test_align_app_reproduces_readme_table(built_lib, pair, golden, tmp_path):
    """apps/align.cpp -- the reference's benchmark program (ndt_omp/apps/align.cpp) over the C-ABI:
    PCD files in, VoxelGrid, three search methods, fitness out.  On the bundled pair it prints the
    fitness values of the reference's README (ndt_omp/README.md:13-46)."""
    from toyslam_amd import ndt
    t, s = pair  # the pair after align.cpp's 0.1 m down-sample (committed fixture)
    tp, sp = str(tmp_path / "target.pcd"), str(tmp_path / "source.pcd")
    ndt.pcd_write_xyz(tp, t)
    ndt.pcd_write_xyz(sp, s)
    exe = str(tmp_path / "align")
    libdir = os.path.join(ROOT, "toyslam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "apps", "align.cpp"),
                           "-o", exe, "-L" + libdir, "-lndt_mi355", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([exe, tp, sp, "0"], text=True)
    gblock = out.split("--- gicp_mi355 ---")[1].split("--- ndt_mi355 (")[0]
    gf = {ln.split(":")[0].strip(): ln.split(":", 1)[1].strip() for ln in gblock.splitlines() if ":" in ln}
    assert gf["converged"].startswith("1") and 0 < float(gf["fitness"]) < 0.25
    blocks = out.split("--- ndt_mi355 (")[1:]
    assert [b.split(")")[0] for b in blocks] == ["KDTREE", "DIRECT7", "DIRECT1"]
    for b in blocks:
        name = b.split(")")[0]
        fields = {ln.split(":")[0].strip(): ln.split(":", 1)[1].strip() for ln in b.splitlines() if ":" in ln}
        assert float(fields["fitness"]) == pytest.approx(golden["readme_fitness"][name], abs=2e-6)
        assert fields["converged"].startswith("1")
        assert float(fields["single"].split("[")[0]) > 0 and float(fields["10times"].split("[")[0]) > 0
    # with the down-sample step (idempotent on an already filtered cloud up to the order of the points)
    out2 = subprocess.check_output([exe, tp, sp, "0.1"], text=True)
    assert "after the 0.10 m voxel grid" in out2


@pytest.mark.gpu
def test_map_sequence_app_follows_the_mapping_node(built_lib, tmp_path):
    """apps/map_sequence.cpp -- the mapping node's processing loop (ndt_omp_mapping_node.cpp:64-211) over the C-ABI:
    numbered PCD scans in, voxel filter, consecutive registrations, pose chain, global map.  Checked against the same
    loop built from the oracle's restatements (VoxelGrid, NDT, transformPointCloud, Eigen's Matrix4f product)."""
    from oracle import pyoracle as po
    from toyslam_amd import clouds, ndt
    rng = np.random.default_rng(9)
    world = clouds.target_surfaces(60000, seed=77, extent=60.0)[:, :3].astype(np.float32)
    steps = [clouds.make_T([0.4, 0.1, 0.0], np.deg2rad([0.0, 0.0, 1.5])), clouds.make_T([0.5, -0.1, 0.02], np.deg2rad([0.2, 0.0, -1.0])),
             clouds.make_T([0.3, 0.2, 0.0], np.deg2rad([0.0, -0.3, 2.0]))]
    pose = np.eye(4)
    scans = []
    for k in range(4):  # the sensor moves through a static world: scan k = the world seen from pose k
        if k:
            pose = pose @ steps[k - 1]
        pick = world[rng.choice(len(world), 30000, replace=False)]
        scans.append((clouds.apply_T(np.linalg.inv(pose), pick) + rng.normal(0, 0.01, pick.shape)).astype(np.float32))
    d = tmp_path / "pcd"
    d.mkdir()
    for k, sc in enumerate(scans, 1):
        ndt.pcd_write_xyz(str(d / ("cloud_%d.pcd" % k)), sc)
    (d / "cloud_3.pcd").rename(d / "cloud_03.pcd")  # numbers, not names, give the order
    exe = str(tmp_path / "map_sequence")
    libdir = os.path.join(ROOT, "toyslam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "apps", "map_sequence.cpp"),
                           "-o", exe, "-L" + libdir, "-lndt_mi355", "-Wl,-rpath," + libdir])
    map_out = str(tmp_path / "map.pcd")
    out = subprocess.check_output([exe, str(d), "0.5", map_out], text=True)
    # "pipeline": the steps of consecutive scans overlap (prep handles, inputs taken over by the registration handle with
    # ndt_share_input_target / _source; with NDT_PIPELINE_PARTITION=1 on CU partitions) -- the same lines, bit for bit
    strip = lambda o: [ln for ln in o.splitlines() if not ln.startswith("time:") and not ln.startswith("start-up") and not ln.startswith("global map written")]
    for env in ({}, {"NDT_PIPELINE_PARTITION": "1"}):
        out_pipe = subprocess.check_output([exe, str(d), "0.5", "-", "node", "pipeline"], text=True, env=dict(os.environ, **env))
        assert strip(out) == strip(out_pipe)
        assert "overlapped with the registrations" in out_pipe and "file reading overlapped" in out
    # the default keeps every cloud in HBM (ndt_cloud); "host" sends them through host buffers as rounds 1-3 did: the same lines
    out_host = subprocess.check_output([exe, str(d), "0.5", "-", "node", "serial", "host"], text=True)
    assert strip(out) == strip(out_host) and "clouds resident in HBM" in out and "clouds through host buffers" in out_host
    lines = out.splitlines()
    traj = []
    for i, ln in enumerate(lines):
        if ln.startswith("trajectory["):
            traj.append(np.array([[float(x) for x in lines[i + 1 + r].split()] for r in range(4)]))
    assert len(traj) == 3 and "clouds 4  registrations 3 (not converged 0)" in out

    # the same loop from the oracle's pieces
    filt = [po.voxel_grid_filter(sc, 0.5)[0] for sc in scans]
    glob = None
    ref_map = po.voxel_grid_filter(filt[0], 0.5)[0]
    for k in range(1, 4):
        o = po.OracleNDT(resolution=1.0, step_size=0.1, trans_eps=0.01, max_iter=64, num_threads=8)
        o.set_target(filt[k - 1])
        o.set_source(filt[k])
        r = o.align()
        assert r["converged"]
        glob = r["T"] if glob is None else ndt.host_chain_pose(glob, r["T"])
        assert rot_err(traj[k - 1], glob) < 2e-4 and trans_err(traj[k - 1], glob) < 2e-3, k  # tolerances add up along the chain
        moved = po.transform_cloud(np.c_[filt[k], np.ones(len(filt[k]), np.float32)], glob)[:, :3]
        ref_map = po.voxel_grid_filter(np.concatenate([ref_map, moved]), 0.5)[0]
        # ... and the registrations roughly recover the motion the scans were generated with (the node's epsilon of 0.01
        # stops the Newton iteration early; independent random subsets of the world per scan)
        assert trans_err(r["T"], steps[k - 1]) < 0.25
    got_map, _ = ndt.pcd_read_xyz(map_out)
    assert abs(len(got_map) - len(ref_map)) <= 0.005 * len(ref_map)  # poses differ in the last bits: a few voxels may flip
    assert "global map %d points" % len(got_map) in out

    # the rosbag node's loop (ndt_rosbag_mapping_node.cpp:46-75,119-141) over the same scans: leaf 0.3, every registration
    # starts from the previous result, fitness printed, pose / trajectory / map updated after every scan
    from scipy.spatial import cKDTree
    out = subprocess.check_output([exe, str(d), "0.3", "-", "rosbag"], text=True)
    assert strip(out) == strip(subprocess.check_output([exe, str(d), "0.3", "-", "rosbag", "pipeline"], text=True))
    assert strip(out) == strip(subprocess.check_output([exe, str(d), "0.3", "-", "rosbag", "serial", "host"], text=True))
    lines = out.splitlines()
    traj = [np.array([[float(x) for x in lines[i + 1 + r].split()] for r in range(4)]) for i, ln in enumerate(lines) if ln.startswith("trajectory[")]
    fit = [float(ln.split()[1]) for ln in lines if ln.startswith("fitness:")]
    assert len(traj) == 3 and len(fit) == 3 and "registrations 3 (not converged 0)" in out
    filt = [po.voxel_grid_filter(sc, 0.3)[0] for sc in scans]
    pose, pres = np.eye(4, dtype=np.float32), None
    for k in range(1, 4):
        o = po.OracleNDT(resolution=1.0, step_size=0.1, trans_eps=0.01, max_iter=64, num_threads=8)
        o.set_target(filt[k - 1])
        o.set_source(filt[k])
        r = o.align(guess=pres, want_cloud=True)
        assert r["converged"]
        d2 = cKDTree(filt[k - 1].astype(np.float64)).query(r["cloud"][:, :3].astype(np.float64))[0] ** 2
        assert abs(fit[k - 1] - d2.mean()) <= 1e-3 * d2.mean(), (k, fit[k - 1], d2.mean())
        pres = r["T"]
        pose = ndt.host_chain_pose(pose, r["T"])
        assert rot_err(traj[k - 1], pose) < 2e-4 and trans_err(traj[k - 1], pose) < 2e-3, k


VGC_SRC = os.path.join(ROOT, "tests", "vgc_harness.cpp")


def test_voxel_grid_covariance_adapter_compiles_and_declares_reference_surface():
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror"] + INC + [VGC_SRC])
    hdr = open(os.path.join(ROOT, "include", "pclomp", "voxel_grid_covariance_omp.h")).read()
    for name in ["class VoxelGridCovariance : public pcl::VoxelGrid<PointT>", "struct Leaf", "typedef std::map<size_t, Leaf> Map",
                 "setMinPointPerVoxel", "getMinPointPerVoxel", "setCovEigValueInflationRatio", "getCovEigValueInflationRatio",
                 "filter(PointCloud& output, bool searchable = false)", "filter(bool searchable = false)", "getLeaf(int index)",
                 "getLeaf(PointT& p)", "getLeaf(Eigen::Vector3f& p)", "getNeighborhoodAtPoint7", "getNeighborhoodAtPoint1",
                 "getLeaves()", "getCentroids()", "nearestKSearch", "radiusSearch", "getEvecs", "getEvals", "getInverseCov",
                 "getPointCount", "voxel_centroids_leaf_indices_", "getDisplayCloud(pcl::PointCloud<pcl::PointXYZ>& cell_cloud)"]:
        assert name in hdr, name


@pytest.mark.gpu
def test_voxel_grid_covariance_adapter_matches_oracle(built_lib, pair, tmp_path):
    """pclomp::VoxelGridCovariance (voxel_grid_covariance_omp.h:59-556) over the C-ABI: leaves, centroid cloud and every query
    against the oracle's leaves and a restatement of the reference's index arithmetic (_impl.hpp:372-444, .h:309-375)."""
    from oracle import pyoracle as po
    t, _ = pair
    exe = str(tmp_path / "vgc_harness")
    libdir = os.path.join(ROOT, "toyslam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + INC + [VGC_SRC, "-o", exe, "-L" + libdir, "-lndt_mi355", "-Wl,-rpath," + libdir])
    tf = str(tmp_path / "t.f32")
    t.astype("<f4").tofile(tf)
    nq = 300
    out = subprocess.check_output([exe, tf, str(len(t)), "1.0", str(nq)], text=True, stderr=subprocess.DEVNULL)
    lines = out.splitlines()
    o = po.OracleNDT(resolution=1.0, num_threads=8)
    o.set_target(t)
    G = o.grid()
    assert lines[0] == "min_points_clamped 3"
    cand = (G["n"] >= 6) | (G["n"] == -1)
    assert lines[1] == "leaves %d output %d centroids %d" % (len(G["idx"]), cand.sum(), cand.sum())
    leaf_lines = [l for l in lines if l.startswith("leaf ")]
    assert len(leaf_lines) == len(G["idx"])
    for i, l in enumerate(leaf_lines):
        f = l.replace("|", " ").split()
        assert int(f[1]) == G["idx"][i] and int(f[2]) == G["n"][i]
        assert np.array_equal(np.array(f[3:6], dtype=np.float64), G["mean"][i])          # means: bit for bit
        if cand[i]:
            c = np.array(f[6:12], dtype=np.float64)
            C = G["cov"][i]
            ref = np.array([C[0, 0], C[0, 1], C[0, 2], C[1, 1], C[1, 2], C[2, 2]])
            assert np.allclose(c, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
            if G["n"][i] >= 6:
                assert np.allclose(np.array(f[12:15], dtype=np.float64), np.diag(G["icov"][i]), rtol=1e-9)
                assert np.allclose(np.array(f[15:18], dtype=np.float64), G["evals"][i], rtol=1e-9, atol=1e-12 * G["evals"][i].max())
        else:
            assert len(f) == 6
    resid, ortho = [float(x) for x in [l for l in lines if l.startswith("eigenbasis")][0].split()[1:]]
    assert resid < 1e-9 and ortho < 1e-12   # C v = lambda v for every valid leaf, V orthonormal
    first = np.flatnonzero(cand)[:5]
    for k, l in enumerate([l for l in lines if l.startswith("centroid ")]):
        assert np.array_equal(np.array(l.split()[2:], dtype=np.float32), G["mean"][first[k]].astype(np.float32))
    # the queries, restated from the reference's index arithmetic over the oracle's leaves
    leaves = {int(ix): i for i, ix in enumerate(G["idx"])}
    mb, xb, db = G["min_b"].astype(np.int64), G["max_b"].astype(np.int64), G["div_b"].astype(np.int64)
    mul = np.array([1, db[0], db[0] * db[1]])
    half = [(-1, -1, -1), (-1, 0, -1), (-1, 1, -1), (0, -1, -1), (0, 0, -1), (0, 1, -1), (1, -1, -1), (1, 0, -1), (1, 1, -1), (-1, -1, 0),
            (0, -1, 0), (1, -1, 0), (-1, 0, 0)]
    off26 = half + [tuple(-a for a in h) for h in half]
    off7 = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    cents = G["mean"][cand].astype(np.float32)
    def neighbours(p, offs):
        ijk = np.floor(p / np.float32(1.0)).astype(np.int64)
        got = []
        for d in offs:
            d = np.array(d)
            if np.all(mb - ijk <= d) and np.all(xb - ijk >= d):
                i = leaves.get(int(((ijk + d - mb) * mul).sum()))
                if i is not None and G["n"][i] >= 6:
                    got.append(i)
        return got
    qlines = [l for l in lines if l.startswith("query ")]
    assert len(qlines) == nq
    for l in qlines:
        f = l.split()
        i = int(f[1])
        p = t[i].astype(np.float32).copy()
        p[0] += np.float32(0.37) * np.float32(i % 5 - 2)
        p[1] -= np.float32(0.21) * np.float32(i % 3 - 1)
        n7 = neighbours(p, off7)
        assert int(f[2]) == len(neighbours(p, off26)) and int(f[3]) == len(n7) and int(f[5]) == len(neighbours(p, off7[:1]))
        assert float(f[4]) == pytest.approx(sum(G["mean"][j][0] * (k + 1) for k, j in enumerate(n7)), rel=1e-14, abs=1e-14)
        c = (np.floor(p * np.float32(1.0)) - mb.astype(np.float32)).astype(np.int64)   # getLeaf: no bounds test (.h:326-346)
        own = leaves.get(int((c * mul).sum()))
        assert int(f[6]) == (G["n"][own] if own is not None else -999)
        d = p[None, :] - cents
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        inside = np.sort(d2[d2 < np.float32(1.0)])
        assert int(f[7]) == len(inside) and float(f[8]) == pytest.approx(float(inside[0]) if len(inside) else -1.0, rel=1e-6)
        srt = np.sort(d2)
        assert int(f[9]) == 3 and float(f[10]) == pytest.approx(float(srt[0]), rel=1e-6) and float(f[11]) == pytest.approx(float(srt[2]), rel=1e-6)
    assert lines[-2] == "copy %d %d" % (len(G["idx"]), len(neighbours(t[0].astype(np.float32), off7)))
    # getDisplayCloud: 1000 draws per voxel with enough points; the first voxel's sample mean within 5 standard errors of its mean
    disp = lines[-1].split()
    n_valid = int((G["n"] >= 6).sum())
    assert disp[0] == "display" and int(disp[1]) == 1000 * n_valid and int(disp[2]) == n_valid and float(disp[3]) < 5.0
