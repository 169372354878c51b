// rccl_sharded_ba.cpp -- local / global bundle adjustment sharded over the GPUs of one node with RCCL, from C++:
// the sharded LM loop of libvo_hip.so (vo_ba_local_ba on a handle restricted to its points by vo_ba_set_shard) with
// its two all-reduces per LM iteration carried by ncclAllReduce(ncclDouble, ncclSum) on the handle's stream -- the
// collective the reference-side caller (Optimizer::solveLocalBAPoseAndPoint, optimizer_ceres.cpp:446-808, one process
// per GPU) would register.  One process per GPU: rank / world size from RANK / WORLD_SIZE (or OMPI_COMM_WORLD_*), the
// RCCL unique id travels through a file (VO_NCCL_ID_FILE, default /tmp/vo_nccl_id; rank 0 writes it) so that no MPI is
// needed:
//
//   for r in 0 1 2 3 4 5 6 7; do RANK=$r WORLD_SIZE=8 ./rccl_sharded_ba problem.bin & done; wait
//
// On a box with ONE GPU (RCCL refuses two ranks on one device) the same program runs as a single rank with
// `--collectives-at-one-rank` (vo_ba_set_option(h, VO_BA_OPT_COLLECTIVES_AT_ONE_RANK, 1)): the library then runs the sharded form of its LM loop on the one shard, so every
// collective of the loop is a real ncclAllReduce on the handle's stream (a sum over one rank) -- what
// tests/test_gpu_rccl.py does.
//
// problem.bin: tools/dump_ba_problem.py (int32 n_cams, n_points, n_edges; poses [n_cams][6] f64; fixed [n_cams] u8;
// points [n_points][3] f64; edge_cam, edge_point [n_edges] i32; edge_obs [n_edges][3] f64; inv_sigma [n_edges] f64;
// cam[5] f64).  Every rank prints its wall time; rank 0 also solves the unsharded problem and reports the largest pose
// difference.  Build (tests/test_rccl_build.py does exactly this, compile + link, in the CPU container):
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 examples/rccl_sharded_ba.cpp -Iinclude -Lvo_slam_test_amd -lvo_hip -lrccl \
//         -Wl,-rpath,$PWD/vo_slam_test_amd -o rccl_sharded_ba
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vo_hip.h"

#define HIP_OK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                              \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)
#define NCCL_OK(x)                                                                         \
  do {                                                                                     \
    ncclResult_t r_ = (x);                                                                 \
    if (r_ != ncclSuccess) {                                                               \
      fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));                             \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)
#define VO_OK_OR_DIE(x)                                                                    \
  do {                                                                                     \
    int s_ = (x);                                                                          \
    if (s_ != VO_OK) {                                                                     \
      fprintf(stderr, "%s: status %d: %s\n", #x, s_, vo_last_error());                     \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

struct Problem {
  int n_cams = 0, n_points = 0, n_edges = 0;
  std::vector<double> poses, points, obs, isg;
  std::vector<uint8_t> fixed;
  std::vector<int32_t> ecam, ept;
  double cam[5];
};

static bool load(const char *path, Problem &P) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  int32_t h[3];
  bool ok = fread(h, 4, 3, f) == 3;
  P.n_cams = h[0], P.n_points = h[1], P.n_edges = h[2];
  P.poses.resize((size_t)6 * P.n_cams), P.fixed.resize(P.n_cams), P.points.resize((size_t)3 * P.n_points);
  P.ecam.resize(P.n_edges), P.ept.resize(P.n_edges), P.obs.resize((size_t)3 * P.n_edges), P.isg.resize(P.n_edges);
  ok = ok && fread(P.poses.data(), 8, P.poses.size(), f) == P.poses.size();
  ok = ok && fread(P.fixed.data(), 1, P.fixed.size(), f) == P.fixed.size();
  ok = ok && fread(P.points.data(), 8, P.points.size(), f) == P.points.size();
  ok = ok && fread(P.ecam.data(), 4, P.ecam.size(), f) == P.ecam.size();
  ok = ok && fread(P.ept.data(), 4, P.ept.size(), f) == P.ept.size();
  ok = ok && fread(P.obs.data(), 8, P.obs.size(), f) == P.obs.size();
  ok = ok && fread(P.isg.data(), 8, P.isg.size(), f) == P.isg.size();
  ok = ok && fread(P.cam, 8, 5, f) == 5;
  fclose(f);
  return ok;
}

// the all-reduce vo_ba's sharded LM loop calls twice per iteration
static int allreduce(void *user, double *dev_buf, size_t n, void *stream) {
  return ncclAllReduce(dev_buf, dev_buf, n, ncclDouble, ncclSum, *static_cast<ncclComm_t *>(user), (hipStream_t)stream) ==
                 ncclSuccess
             ? 0
             : 1;
}

static int env_int(const char *a, const char *b, int dflt) {
  const char *v = getenv(a);
  if (!v) v = getenv(b);
  return v ? atoi(v) : dflt;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s problem.bin [--collectives-at-one-rank] [--segments]\n", argv[0]);
    return 2;
  }
  const int rank = env_int("RANK", "OMPI_COMM_WORLD_RANK", 0), world = env_int("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", 1);
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  HIP_OK(hipSetDevice(env_int("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", rank) % (ndev > 0 ? ndev : 1)));
  Problem P;
  if (!load(argv[1], P)) {
    fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  // RCCL communicator: the unique id through a file
  ncclUniqueId id;
  const char *idf = getenv("VO_NCCL_ID_FILE");
  const std::string id_path = idf ? idf : "/tmp/vo_nccl_id";
  if (rank == 0) {
    NCCL_OK(ncclGetUniqueId(&id));
    const std::string tmp = id_path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(&id, sizeof(id), 1, f) != 1) return 2;
    fclose(f);
    rename(tmp.c_str(), id_path.c_str());
  } else {
    for (int tries = 0;; tries++) {
      FILE *f = fopen(id_path.c_str(), "rb");
      if (f) {
        const bool got = fread(&id, sizeof(id), 1, f) == 1;
        fclose(f);
        if (got) break;
      }
      if (tries > 600) {
        fprintf(stderr, "rank %d: no RCCL id in %s\n", rank, id_path.c_str());
        return 2;
      }
      usleep(100000);
    }
  }
  ncclComm_t comm;
  NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
  hipStream_t st;
  HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  {
    // self-check of the collective on the stream the LM loop will use: every rank contributes (rank + 1) * i
    const int N = 4096;
    std::vector<double> hbuf(N);
    for (int i = 0; i < N; i++) hbuf[i] = (double)(rank + 1) * i;
    double *dbuf = nullptr;
    HIP_OK(hipMalloc(&dbuf, N * sizeof(double)));
    HIP_OK(hipMemcpyAsync(dbuf, hbuf.data(), N * sizeof(double), hipMemcpyHostToDevice, st));
    if (allreduce(&comm, dbuf, N, st) != 0) {
      fprintf(stderr, "ncclAllReduce failed\n");
      return 1;
    }
    HIP_OK(hipMemcpyAsync(hbuf.data(), dbuf, N * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipFree(dbuf));
    const double tri = 0.5 * world * (world + 1);
    for (int i = 0; i < N; i++)
      if (hbuf[i] != tri * i) {
        fprintf(stderr, "rank %d: RCCL all-reduce self-check failed at %d: %g != %g\n", rank, i, hbuf[i], tri * i);
        return 1;
      }
    printf("rank %d / %d: RCCL all-reduce self-check on the BA stream ok (%d doubles)\n", rank, world, N);
  }
  static int n_calls = 0;  // collectives the LM loops asked for
  struct Counting {
    ncclComm_t *comm;
  } counting{&comm};
  auto counted = +[](void *user, double *buf, size_t n, void *stream) -> int {
    n_calls++;
    return allreduce(static_cast<Counting *>(user)->comm, buf, n, stream);
  };

  vo_ba *h = nullptr;
  VO_OK_OR_DIE(vo_ba_create(&h, P.n_cams, P.poses.data(), P.fixed.data(), P.n_points, P.points.data(), P.n_edges, P.ecam.data(),
                            P.ept.data(), P.obs.data(), P.isg.data(), P.cam));
  VO_OK_OR_DIE(vo_ba_set_stream(h, st));
  VO_OK_OR_DIE(vo_ba_set_shard(h, rank, world));
  for (int a = 2; a < argc; a++) {  // protocol options: the same flags on every rank (the library checks it by a handshake)
    if (!strcmp(argv[a], "--collectives-at-one-rank")) VO_OK_OR_DIE(vo_ba_set_option(h, VO_BA_OPT_COLLECTIVES_AT_ONE_RANK, 1));
    if (!strcmp(argv[a], "--segments")) VO_OK_OR_DIE(vo_ba_set_option(h, VO_BA_OPT_SEGMENTS, 1));
  }
  VO_OK_OR_DIE(vo_ba_set_allreduce(h, counted, &counting));
  std::vector<uint8_t> erase(P.n_edges);
  vo_lm_summary sums[2];
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {  // the first repetition carries RCCL's connection set-up
    VO_OK_OR_DIE(vo_ba_set_state(h, P.poses.data(), P.points.data()));
    HIP_OK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    VO_OK_OR_DIE(vo_ba_local_ba(h, nullptr, erase.data(), sums));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms < best) best = ms;
  }
  std::vector<double> poses((size_t)6 * P.n_cams), points((size_t)3 * P.n_points);
  VO_OK_OR_DIE(vo_ba_get_state(h, poses.data(), points.data()));
  const int iters = sums[0].iterations + sums[1].iterations;
  printf("rank %d / %d: %d LM iterations in %.3f ms (%.1f iterations/s), final cost %.6e, %d RCCL all-reduces in 3 solves\n", rank,
         world, iters, best, iters / best * 1e3, sums[1].final_cost, n_calls);
  vo_ba_destroy(h);
  if (rank == 0) {
    vo_ba *u = nullptr;
    VO_OK_OR_DIE(vo_ba_create(&u, P.n_cams, P.poses.data(), P.fixed.data(), P.n_points, P.points.data(), P.n_edges, P.ecam.data(),
                              P.ept.data(), P.obs.data(), P.isg.data(), P.cam));
    std::vector<uint8_t> e2(P.n_edges);
    vo_lm_summary s2[2];
    VO_OK_OR_DIE(vo_ba_local_ba(u, nullptr, e2.data(), s2));
    std::vector<double> p2((size_t)6 * P.n_cams), q2((size_t)3 * P.n_points);
    VO_OK_OR_DIE(vo_ba_get_state(u, p2.data(), q2.data()));
    double worst = 0;
    for (size_t i = 0; i < p2.size(); i++) worst = std::fmax(worst, std::fabs(p2[i] - poses[i]));
    printf("sharded over %d ranks vs one GPU: max |pose difference| %.3e, erase masks %s\n", world, worst,
           memcmp(e2.data(), erase.data(), erase.size()) == 0 ? "identical" : "DIFFER");
    vo_ba_destroy(u);
  }
  NCCL_OK(ncclCommDestroy(comm));
  return 0;
}
