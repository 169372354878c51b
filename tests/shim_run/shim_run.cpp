// shim_run.cpp -- test driver (tests/test_gpu_shims_run.py): the C++ shims of include/myslam_shim/ EXECUTED, not just
// parsed.  Compiled with g++ against the functional stand-ins of this directory (thirdparty/: minimal Eigen / Sophus /
// cv / DBoW3; myslam/types.h: the builder's own re-declaration of the reference classes with bodies) and linked with
// libvo_hip.so; it reads a file of named arrays, fills Frame / KeyFrame / MapPoint / Map objects from them, calls
//   ORB_SLAM2::ORBextractor::operator()          (ORBextractor.h)
//   myslam::Frame::Frame                         (frame_hip.inl)
//   myslam::Matcher::searchByProjection(F*, F*)  (matcher_hip.inl)
//   myslam::Optimizer::solvePoseOnlySE3          (optimizer_hip.inl)
//   myslam::Optimizer::solveLocalBAPoseAndPoint  (optimizer_hip.inl), also with the stop flag raised
// and writes what they left in the objects to a second file.  The Python test compares that with the CPU oracle.
// No reference source is involved: the classes are re-declared by the builder with the member names the shims use.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "myslam/types.h"
namespace myslam {
mutex MapPoint::mutexOptimizer_;
void MapPoint::computeDescriptor() {}
}  // namespace myslam
#include "myslam_shim/optimizer_hip.inl"
#include "myslam_shim/matcher_hip.inl"
#define VO_SHIM_KEYFRAME_BOW 1
#include "myslam_shim/frame_hip.inl"

// ---- named arrays: "VOBN", int32 n, then per array: int32 name length, name, int32 dtype (0 u8, 1 i32, 2 f32, 3 f64),
// int32 ndim, int64 dims[ndim], data
struct Arr {
  int dtype = 0;
  std::vector<long long> dims;
  std::vector<unsigned char> data;
  long long count() const { long long n = 1; for (long long d : dims) n *= d; return n; }
  template <class T> const T *as() const { return reinterpret_cast<const T *>(data.data()); }
};
typedef std::map<std::string, Arr> Blob;
static const int kElem[4] = {1, 4, 4, 8};
static bool read_blob(const char *path, Blob &b) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  char magic[4];
  int32_t n = 0;
  if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "VOBN", 4) || fread(&n, 4, 1, f) != 1) return false;
  for (int i = 0; i < n; i++) {
    int32_t len = 0, dt = 0, nd = 0;
    if (fread(&len, 4, 1, f) != 1) return false;
    std::string name(len, ' ');
    if (fread(&name[0], 1, len, f) != (size_t)len || fread(&dt, 4, 1, f) != 1 || fread(&nd, 4, 1, f) != 1) return false;
    Arr a;
    a.dtype = dt, a.dims.resize(nd);
    if (nd && fread(a.dims.data(), 8, nd, f) != (size_t)nd) return false;
    a.data.resize((size_t)a.count() * kElem[dt]);
    if (!a.data.empty() && fread(a.data.data(), 1, a.data.size(), f) != a.data.size()) return false;
    b[name] = a;
  }
  fclose(f);
  return true;
}
static void put(Blob &b, const std::string &name, int dtype, std::vector<long long> dims, const void *p) {
  Arr a;
  a.dtype = dtype, a.dims = dims;
  a.data.resize((size_t)a.count() * kElem[dtype]);
  if (!a.data.empty()) memcpy(a.data.data(), p, a.data.size());
  b[name] = a;
}
static bool write_blob(const char *path, const Blob &b) {
  FILE *f = fopen(path, "wb");
  if (!f) return false;
  const int32_t n = (int32_t)b.size();
  fwrite("VOBN", 1, 4, f), fwrite(&n, 4, 1, f);
  for (const auto &kv : b) {
    const int32_t len = (int32_t)kv.first.size(), dt = kv.second.dtype, nd = (int32_t)kv.second.dims.size();
    fwrite(&len, 4, 1, f), fwrite(kv.first.data(), 1, len, f), fwrite(&dt, 4, 1, f), fwrite(&nd, 4, 1, f);
    if (nd) fwrite(kv.second.dims.data(), 8, nd, f);
    if (!kv.second.data.empty()) fwrite(kv.second.data.data(), 1, kv.second.data.size(), f);
  }
  fclose(f);
  return true;
}

using namespace myslam;

static Camera make_camera(const double *cam5, float baseline) {
  Camera c;
  c.fx_ = (float)cam5[0], c.fy_ = (float)cam5[1], c.cx_ = (float)cam5[2], c.cy_ = (float)cam5[3], c.bf_ = (float)cam5[4];
  c.b_ = baseline;
  c.xMin_ = 0, c.xMax_ = 640, c.yMin_ = 0, c.yMax_ = 480;
  c.gridPerPixelWidth_ = 64.f / 640.f, c.gridPerPixelHeight_ = 48.f / 480.f;
  return c;
}
static SE3 se3_from(const double *xi) {
  Eigen::Map<const Eigen::Matrix<double, 6, 1>> v(xi);
  return SE3::exp(v);
}
static void frame_arrays(Blob &out, const std::string &pre, Frame &F) {
  const int n = (int)F.N_;
  std::vector<float> kp((size_t)n * 7), un((size_t)n * 2);
  std::vector<int32_t> oct(n);
  for (int i = 0; i < n; i++) {
    const cv::KeyPoint &k = F.keypoints_[i];
    kp[7 * i] = k.pt.x, kp[7 * i + 1] = k.pt.y, kp[7 * i + 2] = k.size, kp[7 * i + 3] = k.angle, kp[7 * i + 4] = k.response;
    kp[7 * i + 5] = (float)k.octave, kp[7 * i + 6] = (float)k.class_id;
    un[2 * i] = F.unKeypoints_[i].pt.x, un[2 * i + 1] = F.unKeypoints_[i].pt.y, oct[i] = k.octave;
  }
  put(out, pre + "kp", 2, {n, 7}, kp.data());
  put(out, pre + "un", 2, {n, 2}, un.data());
  put(out, pre + "uright", 2, {n}, F.uRight_.data());
  put(out, pre + "depth", 2, {n}, F.depth_.data());
  put(out, pre + "desc", 0, {n, 32}, n ? F.descriptors_.data : nullptr);
  std::vector<int32_t> cells;  // grid cell sizes, ix-major like gridKeypoints_
  for (int ix = 0; ix < FRAME_GRID_COLS; ix++)
    for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) cells.push_back((int32_t)F.gridKeypoints_[ix][iy].size());
  put(out, pre + "grid_counts", 1, {FRAME_GRID_COLS * FRAME_GRID_ROWS}, cells.data());
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
    return 2;
  }
  Blob in, out;
  if (!read_blob(argv[1], in)) {
    fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  const double *cam5 = in["cam"].as<double>();
  Camera camera = make_camera(cam5, 0.08f);
  camera.distCoef_.create(5, 1, CV_32F);
  for (int i = 0; i < 5; i++) camera.distCoef_.at<float>(i) = in["dist"].as<float>()[i];

  // ---- 1: ORBextractor::operator() and Frame::Frame on a gray image + metric depth image
  ORB_SLAM2::ORBextractor orb(1000, 1.2f, 8, 20, 7);
  const Arr &img = in["image"], &dep = in["depth"];
  const int H = (int)img.dims[0], W = (int)img.dims[1];
  Mat gray(H, W, CV_8UC1, (void *)img.data.data()), depth(H, W, CV_32F, (void *)dep.data.data());
  {
    std::vector<cv::KeyPoint> kps;
    Mat desc;
    orb(gray, Mat(), kps, desc);
    std::vector<float> kp((size_t)kps.size() * 7);
    for (size_t i = 0; i < kps.size(); i++) {
      const cv::KeyPoint &k = kps[i];
      kp[7 * i] = k.pt.x, kp[7 * i + 1] = k.pt.y, kp[7 * i + 2] = k.size, kp[7 * i + 3] = k.angle, kp[7 * i + 4] = k.response;
      kp[7 * i + 5] = (float)k.octave, kp[7 * i + 6] = (float)k.class_id;
    }
    put(out, "orb_kp", 2, {(long long)kps.size(), 7}, kp.data());
    put(out, "orb_desc", 0, {(long long)kps.size(), 32}, kps.empty() ? nullptr : desc.data);
    std::vector<float> sf = orb.GetScaleFactors();
    put(out, "orb_scale", 2, {(long long)sf.size()}, sf.data());
    Mat empty_img;
    std::vector<cv::KeyPoint> none(3);
    Mat dnone(1, 32, CV_8U);
    orb(empty_img, Mat(), none, dnone);  // `if(_image.empty()) return;` -- outputs untouched
    const int32_t untouched = none.size() == 3 && dnone.rows == 1;
    put(out, "orb_empty_untouched", 1, {1}, &untouched);
  }
  Frame cur(gray, depth, "0.0", &camera, &orb);
  frame_arrays(out, "frame_", cur);

  // ---- 2: Matcher::searchByProjection(Frame*, Frame*): `last` carries one map point per feature of `cur`, at the 3-D
  // positions the test chose; cur at the identity (so the projection is exactly what the test computes), last 0.2 m behind it
  {
    const Arr &P = in["match_points"], &obsc = in["match_obs_cnt"], &lout = in["match_last_outlier"];
    const int n = (int)P.dims[0];
    Frame last;
    last.camera_ = &camera, last.scaleFactors_ = cur.scaleFactors_, last.N_ = n;
    last.unKeypoints_.assign(cur.unKeypoints_.begin(), cur.unKeypoints_.begin() + n);
    last.keypoints_ = last.unKeypoints_;
    last.outliers_.assign(n, false);
    std::vector<MapPoint> mps(n);
    last.mappoints_.assign(n, nullptr);
    for (int i = 0; i < n; i++) {
      mps[i].pos_ = Vector3d(P.as<double>()[3 * i], P.as<double>()[3 * i + 1], P.as<double>()[3 * i + 2]);
      mps[i].descriptor_ = cur.descriptors_.row(i).clone();
      mps[i].observe_cnt_ = obsc.as<int32_t>()[i];
      if (P.as<double>()[3 * i + 2] != 0.0) last.mappoints_[i] = &mps[i];  // z == 0 marks "no map point"
      last.outliers_[i] = lout.as<uint8_t>()[i] != 0;
    }
    const double back[6] = {0, 0, 0.2, 0, 0, 0};
    last.setPose(se3_from(back));
    cur.setPose(SE3());
    Matcher matcher(0.8f);
    const int nm = matcher.searchByProjection(&cur, &last, 15.0f, true);
    std::vector<int32_t> assigned(cur.N_, -1);
    for (size_t k = 0; k < cur.N_; k++)
      if (cur.mappoints_[k]) assigned[k] = (int32_t)(cur.mappoints_[k] - mps.data());
    const int32_t nm32 = nm;
    put(out, "match_n", 1, {1}, &nm32);
    put(out, "match_assigned", 1, {(long long)cur.N_}, assigned.data());
    cur.mappoints_.assign(cur.N_, nullptr);
  }

  // ---- 3: Optimizer::solvePoseOnlySE3(Frame*): a frame whose features are the problem's observations; every 7th
  // feature has no map point (the gather must skip it and leave its outlier flag alone)
  {
    const Arr &pts = in["pose_pts"], &obs = in["pose_obs"], &oct = in["pose_octave"];
    const int n = (int)pts.dims[0];
    Frame F;
    F.camera_ = &camera, F.scaleFactors_ = orb.GetScaleFactors(), F.N_ = n;
    F.unKeypoints_.resize(n), F.uRight_.resize(n), F.mappoints_.assign(n, nullptr), F.outliers_.assign(n, true);
    std::vector<MapPoint> mps(n);
    for (int i = 0; i < n; i++) {
      F.unKeypoints_[i].pt.x = (float)obs.as<double>()[3 * i], F.unKeypoints_[i].pt.y = (float)obs.as<double>()[3 * i + 1];
      F.unKeypoints_[i].octave = oct.as<int32_t>()[i];
      F.uRight_[i] = (float)obs.as<double>()[3 * i + 2];
      mps[i].pos_ = Vector3d(pts.as<double>()[3 * i], pts.as<double>()[3 * i + 1], pts.as<double>()[3 * i + 2]);
      if (i % 7 != 3) F.mappoints_[i] = &mps[i];
    }
    F.setPose(se3_from(in["pose_pose0"].as<double>()));
    const int32_t inl = Optimizer::solvePoseOnlySE3(&F);
    put(out, "pose_inliers", 1, {1}, &inl);
    std::vector<uint8_t> o(n);
    for (int i = 0; i < n; i++) o[i] = F.outliers_[i];
    put(out, "pose_outliers", 0, {n}, o.data());
    const Eigen::Matrix<double, 6, 1> xi = F.Tcw_.log();
    put(out, "pose_pose", 3, {6}, xi.data());
    // no observations at all: 0 inliers, pose untouched (:204-205)
    Frame E;
    E.camera_ = &camera, E.scaleFactors_ = F.scaleFactors_, E.mappoints_.assign(4, nullptr), E.unKeypoints_.resize(4), E.uRight_.assign(4, -1.f);
    E.outliers_.assign(4, false);
    const int32_t inl0 = Optimizer::solvePoseOnlySE3(&E);
    put(out, "pose_empty_inliers", 1, {1}, &inl0);
  }

  // ---- 4: Optimizer::solveLocalBAPoseAndPoint(KeyFrame*, bool&, Map*): key-frames / map points built from the edge list
  {
    const Arr &poses = in["lba_poses"], &points = in["lba_points"], &ecam = in["lba_e_cam"], &ept = in["lba_e_pt"],
              &eobs = in["lba_e_obs"], &eoct = in["lba_e_octave"], &efeat = in["lba_e_feat"];
    const int nc = (int)poses.dims[0], np = (int)points.dims[0], ne = (int)ecam.dims[0];
    const int n_local = in["lba_n_local"].as<int32_t>()[0];  // key-frames 0 .. n_local - 1 are the local window, the rest fixed
    for (int pass = 0; pass < 2; pass++) {  // pass 0: stop flag raised (no write-back); pass 1: the real solve
      std::vector<KeyFrame> kfs(nc);
      std::vector<MapPoint> mps(np);
      std::vector<int> nfeat(nc, 0);
      for (int e = 0; e < ne; e++) nfeat[ecam.as<int32_t>()[e]] = std::max(nfeat[ecam.as<int32_t>()[e]], efeat.as<int32_t>()[e] + 1);
      for (int c = 0; c < nc; c++) {
        kfs[c].id_ = (unsigned long)c, kfs[c].camera_ = &camera, kfs[c].scaleFactors_ = orb.GetScaleFactors();
        kfs[c].Tcw_ = se3_from(poses.as<double>() + 6 * c);
        kfs[c].unKeypoints_.resize(nfeat[c]), kfs[c].uRight_.assign(nfeat[c], -1.f), kfs[c].mappoints_.assign(nfeat[c], nullptr);
      }
      for (int j = 0; j < np; j++) mps[j].pos_ = Vector3d(points.as<double>()[3 * j], points.as<double>()[3 * j + 1], points.as<double>()[3 * j + 2]);
      for (int e = 0; e < ne; e++) {
        KeyFrame &kf = kfs[ecam.as<int32_t>()[e]];
        MapPoint &mp = mps[ept.as<int32_t>()[e]];
        const int ft = efeat.as<int32_t>()[e];
        kf.unKeypoints_[ft].pt.x = (float)eobs.as<double>()[3 * e], kf.unKeypoints_[ft].pt.y = (float)eobs.as<double>()[3 * e + 1];
        kf.unKeypoints_[ft].octave = eoct.as<int32_t>()[e];
        kf.uRight_[ft] = (float)eobs.as<double>()[3 * e + 2];
        kf.mappoints_[ft] = &mp;
        mp.addObservation(&kf, (size_t)ft);
        mp.observe_cnt_ += 10;  // (keep eraseObservedKF from flagging points bad: that bookkeeping is the map's)
      }
      KeyFrame *cur_kf = &kfs[n_local - 1];
      for (int c = 0; c < n_local - 1; c++) cur_kf->ordered_.push_back(&kfs[c]);
      Map map;
      bool stop = pass == 0;
      Optimizer::solveLocalBAPoseAndPoint(cur_kf, stop, &map);
      const std::string pre = pass == 0 ? "lba_stopped_" : "lba_";
      std::vector<double> po((size_t)6 * nc), pt((size_t)3 * np);
      std::vector<int32_t> calls(nc), nupd(np), fixid(nc);
      for (int c = 0; c < nc; c++) {
        const Eigen::Matrix<double, 6, 1> xi = kfs[c].Tcw_.log();
        memcpy(&po[6 * c], xi.data(), 48);
        calls[c] = kfs[c].set_pose_calls_, fixid[c] = kfs[c].BAFixId_ == cur_kf->id_;
      }
      for (int j = 0; j < np; j++) {
        memcpy(&pt[3 * j], mps[j].pos_.data(), 24);
        nupd[j] = mps[j].normal_updates_;
      }
      std::vector<uint8_t> still(ne), slot(ne);
      for (int e = 0; e < ne; e++) {
        KeyFrame &kf = kfs[ecam.as<int32_t>()[e]];
        MapPoint &mp = mps[ept.as<int32_t>()[e]];
        still[e] = mp.observedKFs_.count(&kf) != 0;
        slot[e] = kf.mappoints_[efeat.as<int32_t>()[e]] != nullptr;
      }
      put(out, pre + "poses", 3, {nc, 6}, po.data());
      put(out, pre + "points", 3, {np, 3}, pt.data());
      put(out, pre + "set_pose_calls", 1, {nc}, calls.data());
      put(out, pre + "normal_updates", 1, {np}, nupd.data());
      put(out, pre + "ba_fix_id", 1, {nc}, fixid.data());
      put(out, pre + "still_observed", 0, {ne}, still.data());
      put(out, pre + "slot_kept", 0, {ne}, slot.data());
    }
  }
  if (!write_blob(argv[2], out)) {
    fprintf(stderr, "cannot write %s\n", argv[2]);
    return 2;
  }
  printf("shim_run: ok\n");
  return 0;
}
