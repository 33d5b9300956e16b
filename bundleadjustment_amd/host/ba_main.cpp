// ba_main.cpp -- command-line driver with the flow of the reference's main() for one
// adjustment (src/main.cpp:3040-7135 without the machine-learning outer loop):
//   read .pho/.eop/.iop/.xyz -> build the problem -> solve on the GPU -> sigma0 -> *.jck files.
// The reference takes its file names and fixed/free choices from #defines (src/main.cpp:467-479,
// 4487-4504); here they are arguments.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <atomic>
#include <chrono>
#include <filesystem>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "ba_gpu.h"
#include "ba_host.h"

// What rank 0 leaves at the head of the rendezvous file (--id-file), so that the other ranks can tell this run's file
// from one that an earlier run left behind: the run id that all ranks were started with (--run-id; 0 when not given)
// and the time rank 0 started.  A rank accepts a file whose run id is its own and, when no run id was given, whose
// stamp is not older than two minutes before its own start.
struct RunStamp {
  uint64_t magic, run_id;
  int64_t started;  // seconds since the epoch
};
static const uint64_t kStampMagic = 0x42414d41494e3033ULL;  // "BAMAIN03"
static int64_t g_started = 0;
static std::string g_cleanup_path;  // rank 0 removes the rendezvous file when it ends
static void remove_rendezvous() {
  if (!g_cleanup_path.empty()) {
    unlink(g_cleanup_path.c_str());
    unlink((g_cleanup_path + ".tmp").c_str());
  }
}
static bool stamp_ok(const RunStamp &st, uint64_t run_id) {
  if (st.magic != kStampMagic || st.run_id != run_id) return false;
  return run_id != 0 || st.started >= g_started - 120;
}

// Reduction over the ranks through a file that all of them map (--comm shm): for machines with fewer GPUs than ranks
// (RCCL wants one GPU per rank) and for tests.  Every rank copies its chunk into its own slot, waits for the others, and
// adds the slots up in rank order -- the same sum in the same order everywhere, so the ranks stay bit-identical.
// A rank that waits longer than `timeout_s` for the others gives up: the reduction fails, and so does the solve.
struct ShmComm {
  static constexpr int64_t kChunk = 1 << 16;
  struct Header {
    RunStamp stamp;
    std::atomic<int> arrived;
    std::atomic<int> generation;
  };
  Header *h = nullptr;
  double *slots = nullptr;
  int ranks = 1, rank = 0;
  double timeout_s = 600.0;
  std::vector<double> host;

  static size_t bytes(int ranks) { return 4096 + sizeof(double) * (size_t)kChunk * ranks; }
  bool open(const std::string &path, int ranks_, int rank_, uint64_t run_id) {
    ranks = ranks_; rank = rank_;
    if (rank == 0) {
      // a fresh file under a fresh inode: what an earlier run left at this path cannot be mistaken for it
      unlink(path.c_str());
      const std::string tmp = path + ".tmp";
      unlink(tmp.c_str());
      const int fd = ::open(tmp.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)bytes(ranks)) != 0) return false;
      void *m = mmap(nullptr, bytes(ranks), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (m == MAP_FAILED) return false;
      h = static_cast<Header *>(m);  // a new file is all zero: nobody has arrived, generation 0
      h->stamp = RunStamp{kStampMagic, run_id, g_started};
      if (rename(tmp.c_str(), path.c_str()) != 0) return false;
      g_cleanup_path = path;
      atexit(remove_rendezvous);
    } else {
      // wait for this run's file: one with another stamp is an earlier run's (rank 0 replaces it when it gets there)
      for (int tries = 0; tries < 1200; ++tries, usleep(100000)) {
        const int fd = ::open(path.c_str(), O_RDWR);
        if (fd < 0) continue;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes(ranks)) { close(fd); continue; }
        void *m = mmap(nullptr, bytes(ranks), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) continue;
        if (stamp_ok(static_cast<Header *>(m)->stamp, run_id)) { h = static_cast<Header *>(m); break; }
        munmap(m, bytes(ranks));
      }
      if (!h) return false;
    }
    slots = reinterpret_cast<double *>(reinterpret_cast<char *>(h) + 4096);
    host.resize(kChunk);
    return true;
  }
  bool barrier() {
    const int gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == ranks) {
      h->arrived.store(0);
      h->generation.fetch_add(1);
      return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load() == gen) {
      usleep(50);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
        fprintf(stderr, "error: rank %d waited %.0f s for the other ranks at a reduction: giving up\n", rank, timeout_s);
        return false;
      }
    }
    return true;
  }
  static int reduce(void *user, double *dev, int64_t count, int32_t op, void *stream) {
    ShmComm *c = static_cast<ShmComm *>(user);
    for (int64_t o = 0; o < count; o += kChunk) {
      const int64_t n = std::min<int64_t>(kChunk, count - o);
      if (ba_copy_to_host(c->slots + (size_t)c->rank * kChunk, dev + o, n, stream)) return -1;
      if (!c->barrier()) return -1;
      for (int64_t k = 0; k < n; ++k) {
        double v = c->slots[k];
        for (int r = 1; r < c->ranks; ++r) {
          const double w = c->slots[(size_t)r * kChunk + k];
          v = op == 1 ? std::max(v, w) : v + w;
        }
        c->host[k] = v;
      }
      if (!c->barrier()) return -1;  // every rank has read the slots before any of them writes the next chunk
      if (ba_copy_to_device(dev + o, c->host.data(), n, stream)) return -1;
    }
    return 0;
  }
};

static void usage() {
  fprintf(stderr,
          "usage: ba_main <file.pho> <file.eop> <file.iop> <file.xyz> [--out DIR] [--free-ap MASK] [--free-iop MASK]\n"
          "               [--solver auto|dense|sparse|pcg|implicit] [--pcg-rtol X] [--max-iter N] [--covariance] [--correlations] [--model M]\n"
          "               [--bins B] [--truth file.xyz] [--iop-bounds]\n"
          "       ba_main --synth C S P RAYS [same options]\n"
          "       ba_main ... --ranks N --rank R --id-file PATH [--run-id ID] [--comm rccl|shm] [--comm-timeout SECONDS]\n"
          "                                          [--pcg-rows auto|never|always]\n"
          "                                          one process per GPU: start N of these, R = 0..N-1\n"
          "  MASK: bit k of --free-ap frees AP[k] (a1 a2 k1 k2 k3 p1 p2 ep1..ep9); default 0x4 = k1 only,\n"
          "        the reference's choice (src/main.cpp:4487-4504).  --free-iop: bits xp yp c, default 7.\n"
          "  --model: projection functor, 0 collinearity (default), 1 collinearity stereographic, 2 fisheye equidistant,\n"
          "        3 equisolid angle, 4 orthographic, 5 stereographic, 6 collinearity orthographic, 7 omnidirectional\n"
          "        (src/main.cpp:3845-3918, :4016, :4186-4250).\n"
          "  --bins: quantile bins of the residual statistics printed after the solve (default 30 = the reference's\n"
          "        QUANTILE_RESIDUALS_BINS, src/main.cpp:51; 0 leaves them out).  --truth: object-space QC against a file\n"
          "        of rows \"id X Y Z sX sY sZ\" (src/main.cpp:7242-7346).\n"
          "  --iop-bounds: the box on xp, yp, c from the sensor format columns of the .iop file, as the reference's\n"
          "        stereographic and fisheye branches set it (src/main.cpp:4252-4256); its collinearity branches set none.\n"
          "  --ranks/--rank: rank R takes its share of the points (contiguous ranges balanced by observation count, all\n"
          "        observations of those points; stations and cameras on every rank), uses GPU R modulo the GPUs that are\n"
          "        visible, and the ranks sum the reduced camera system over RCCL.  Rank 0 writes the communicator's id to\n"
          "        PATH, the others wait for it.  Every rank writes the results of its own points to DIR/rank<R>.\n"
          "        --comm shm sums through a memory-mapped file at PATH instead (ranks on one machine that share GPUs).\n"
          "        --run-id: a number (not 0) that the ranks of one run share and no earlier run had, e.g. the launcher's pid:\n"
          "        rank 0 stamps the file at PATH with it and the others take no file with another stamp (without it they take\n"
          "        any file that rank 0 of some run stamped in the last two minutes).  Rank 0 replaces whatever is at PATH and\n"
          "        removes the file when it ends.  --comm-timeout (default 600): a rank that waits that long for the others\n"
          "        at a reduction, or whose solve has not come back after ten times that, ends with an error.\n"
          "        --pcg-rows: whether the ranks split the rows of the product inside the block-sparse PCG (one more sum per\n"
          "        PCG iteration); auto: when the network is large enough for that to pay.\n"
          "  --covariance: standard deviations of all parameters in the result files (the reference's COMPUTECX\n"
          "        branch, src/main.cpp:5261): from the dense inverse of the reduced system while it fits the device, from a\n"
          "        dissection of the station graph beyond (networks of 50 000 stations take seconds).\n"
          "  --correlations: --covariance and the correlation tables of the reference's COMPUTECORRELATION branch\n"
          "        (src/main.cpp:5592-6495): camera parameters against station parameters, point coordinates and each other.\n");
}

int main(int argc, char **argv) {
  std::vector<std::string> pos;
  std::string out = ".";
  unsigned ap_mask = 0x4, iop_mask = 7;
  bool synth = false, covariance = false, correlations = false, iop_bounds = false;
  int model = 0, bins = 30;
  std::string truth, id_file, comm = "rccl", pcg_rows = "auto";
  uint64_t run_id = 0;
  double comm_timeout = 600.0;
  g_started = (int64_t)time(nullptr);
  ShmComm shm;
  int ranks = 1, rank = 0;
  ba_options opt;
  ba_default_options(&opt);
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "--out") out = next();
    else if (a == "--free-ap") ap_mask = (unsigned)strtoul(next(), nullptr, 0);
    else if (a == "--free-iop") iop_mask = (unsigned)strtoul(next(), nullptr, 0);
    else if (a == "--solver") {
      const std::string v = next();
      opt.linear_solver = v == "pcg" ? BA_SOLVER_PCG_SCHUR : v == "implicit" ? BA_SOLVER_PCG_IMPLICIT : v == "dense" ? BA_SOLVER_DENSE_SCHUR :
                          v == "sparse" ? BA_SOLVER_SPARSE_SCHUR : BA_SOLVER_AUTO;
    }
    else if (a == "--pcg-rtol") opt.pcg_rtol = atof(next());
    else if (a == "--max-iter") opt.max_num_iterations = atoi(next());
    else if (a == "--synth") synth = true;
    else if (a == "--covariance") covariance = true;
    else if (a == "--correlations") covariance = correlations = true;
    else if (a == "--iop-bounds") iop_bounds = true;
    else if (a == "--model") model = atoi(next());
    else if (a == "--bins") bins = atoi(next());
    else if (a == "--truth") truth = next();
    else if (a == "--ranks") ranks = atoi(next());
    else if (a == "--rank") rank = atoi(next());
    else if (a == "--id-file") id_file = next();
    else if (a == "--comm") comm = next();
    else if (a == "--run-id") run_id = strtoull(next(), nullptr, 0);
    else if (a == "--comm-timeout") comm_timeout = atof(next());
    else if (a == "--pcg-rows") pcg_rows = next();
    else pos.push_back(a);
  }
  ba_network *net = nullptr;
  if (synth) {
    if (pos.size() != 4) { usage(); return 2; }
    ba_synth_config cfg;
    ba_synth_default(&cfg);
    cfg.n_cam = atoi(pos[0].c_str()); cfg.n_sta = atoi(pos[1].c_str()); cfg.n_pts = atoi(pos[2].c_str());
    cfg.rays_per_point = atoi(pos[3].c_str());
    cfg.ap_free_mask = (uint16_t)ap_mask; cfg.iop_free_mask = (uint8_t)iop_mask;
    cfg.model = model;
    if (ba_net_synth(&cfg, &net)) { fprintf(stderr, "error: %s\n", ba_host_last_error()); return 1; }
  } else {
    if (pos.size() != 4) { usage(); return 2; }
    if (ba_net_read(pos[0].c_str(), pos[1].c_str(), pos[2].c_str(), pos[3].c_str(), &net)) {
      fprintf(stderr, "error: %s\n", ba_host_last_error());
      return 1;
    }
    ba_net_set_free(net, (uint16_t)ap_mask, (uint8_t)iop_mask);
  }
  if (ba_net_set_model(net, model)) { fprintf(stderr, "error: %s\n", ba_host_last_error()); return 1; }
  if (ranks < 1 || rank < 0 || rank >= ranks || (ranks > 1 && id_file.empty())) { usage(); return 2; }
  if (ranks > 1) {
    if (covariance) { fprintf(stderr, "error: --covariance is for one rank\n"); return 2; }
    // the output directory first: nothing below this point may start another program (the GPU is initialised from here on)
    out += "/rank" + std::to_string(rank);
    std::error_code ec;
    std::filesystem::create_directories(out, ec);
    if (ec) { fprintf(stderr, "error: cannot create %s: %s\n", out.c_str(), ec.message().c_str()); return 1; }
    // this rank's points, all of their observations; the whole network is let go
    ba_network *shard = nullptr;
    int64_t lo = 0, hi = 0;
    if (ba_net_shard(net, rank, ranks, &shard, &lo, &hi)) { fprintf(stderr, "error: %s\n", ba_host_last_error()); return 1; }
    ba_net_free(net);
    net = shard;
    printf("    Rank %d of %d: points [%lld, %lld)\n", rank, ranks, (long long)lo, (long long)hi);
    int32_t n_dev = 0;
    ba_device_count(&n_dev);
    if (n_dev < 1 || ba_set_device(rank % n_dev)) { fprintf(stderr, "error: no GPU for rank %d: %s\n", rank, ba_last_error()); return 1; }
  }
  ba_dims d;
  ba_inputs in;
  ba_net_dims(net, &d);
  ba_net_inputs(net, &in);
  printf("    Number of observations read: %d\n    Number of EOPs read: %d\n    Number of IOPs read: %d\n    Number of XYZ read: %d\n",
         d.n_obs, d.n_sta, d.n_cam, d.n_pts);
  ba_problem *p = nullptr;
  if (ba_create(&d, &in, &p)) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
  if (ranks > 1 && comm == "shm") {
    shm.timeout_s = comm_timeout;
    if (!shm.open(id_file, ranks, rank, run_id)) { fprintf(stderr, "error: cannot map %s (no file of this run appeared there)\n", id_file.c_str()); return 1; }
    if (ba_set_allreduce(p, ShmComm::reduce, &shm)) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
  } else if (ranks > 1) {
    // the id of the communicator travels through a file: written under another name and renamed, so that a reader
    // never sees half of it
    unsigned char id[128];
    if (rank == 0) {
      if (ba_rccl_unique_id(id)) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
      unlink(id_file.c_str());  // an earlier run's file
      const std::string tmp = id_file + ".tmp";
      const RunStamp st{kStampMagic, run_id, g_started};
      FILE *f = fopen(tmp.c_str(), "wb");
      if (!f || fwrite(&st, sizeof st, 1, f) != 1 || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 ||
          rename(tmp.c_str(), id_file.c_str()) != 0) {
        fprintf(stderr, "error: cannot write %s\n", id_file.c_str());
        return 1;
      }
      g_cleanup_path = id_file;
      atexit(remove_rendezvous);
    } else {
      bool have = false;
      for (int tries = 0; tries < 1200 && !have; ++tries) {
        RunStamp st{};
        FILE *f = fopen(id_file.c_str(), "rb");
        if (f) {
          have = fread(&st, sizeof st, 1, f) == 1 && fread(id, 1, sizeof id, f) == sizeof id && stamp_ok(st, run_id);
          fclose(f);
        }
        if (!have) usleep(100000);
      }
      if (!have) { fprintf(stderr, "error: no communicator id of this run in %s\n", id_file.c_str()); return 1; }
    }
    if (ba_use_rccl(p, ranks, rank, id)) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
  }
  if (ranks > 1) {
    const int mode = pcg_rows == "never" ? -1 : pcg_rows == "always" ? 1 : 0;
    if (ba_shard_reduced_solve(p, rank, ranks, mode)) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
  }
  if (iop_bounds) {
    std::vector<double> lo(3 * (size_t)d.n_cam), hi(3 * (size_t)d.n_cam);
    ba_net_iop_bounds(net, lo.data(), hi.data());
    if (ba_set_iop_bounds(p, lo.data(), hi.data())) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
  }
  ba_summary sm;
  std::vector<ba_iter_row> rows(opt.max_num_iterations + 2);
  // Several ranks: a rank that another one has left alone inside a collective would wait for ever.  A watchdog ends the
  // process with a message instead when the solve does not come back.
  std::atomic<bool> solve_done{false};
  std::thread watchdog;
  if (ranks > 1) {
    watchdog = std::thread([&]() {
      const auto t0 = std::chrono::steady_clock::now();
      while (!solve_done.load()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0 * comm_timeout) {
          fprintf(stderr, "error: rank %d: the solve has not returned after %.0f s (another rank gone?): giving up\n", rank, 10.0 * comm_timeout);
          remove_rendezvous();
          _exit(3);
        }
      }
    });
  }
  struct JoinWatchdog {
    std::atomic<bool> &done;
    std::thread &t;
    ~JoinWatchdog() {
      done.store(true);
      if (t.joinable()) t.join();
    }
  } join_watchdog{solve_done, watchdog};
  if (ba_solve(p, &opt, ba_net_eop(net), ba_net_xyz(net), ba_net_iop(net), ba_net_ap(net), &sm, rows.data(), (int32_t)rows.size())) {
    fprintf(stderr, "error: %s\n", ba_last_error());
    return 1;
  }
  solve_done.store(true);
  // the table ceres prints with minimizer_progress_to_stdout (src/main.cpp:4711)
  if (rank == 0) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter\n");
  for (int i = 0; rank == 0 && i < sm.num_iterations && i < (int)rows.size(); ++i)
    printf("%4d  %.6e  %9.2e  %9.2e  %9.2e  %9.2e  %9.2e  %6d\n", rows[i].iteration, rows[i].cost, rows[i].cost_change,
           rows[i].gradient_max_norm, rows[i].step_norm, rows[i].relative_decrease, rows[i].trust_region_radius,
           rows[i].linear_iterations);
  if (rank == 0) printf("Initial cost %.6e  Final cost %.6e  Termination %s\n", sm.initial_cost, sm.final_cost,
         sm.termination == BA_CONVERGENCE ? "CONVERGENCE" : sm.termination == BA_NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE");
  if (covariance) {
    std::vector<double> ve(6 * (size_t)d.n_sta + 1), vx(3 * (size_t)d.n_pts + 1), vi(3 * (size_t)d.n_cam + 1), va(16 * (size_t)d.n_cam + 1);
    const int rc = ba_covariance(p, ve.data(), vx.data(), vi.data(), va.data());
    if (rc == BA_ERR_UNSUPPORTED) {
      // neither route applies (no block-sparse reduced system, or a station graph without separators): the camera
      // parameters only, the other files keep the default sigma
      printf("  %s: standard deviations of the camera parameters only\n", ba_last_error());
      if (ba_covariance_cameras(p, vi.data(), va.data())) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
      ba_net_set_variances(net, nullptr, nullptr, vi.data(), va.data());
    } else if (rc) {
      fprintf(stderr, "error: %s\n", ba_last_error());
      return 1;
    } else {
      ba_net_set_variances(net, ve.data(), vx.data(), vi.data(), va.data());
      if (correlations) {
        // the reference's COMPUTECORRELATION block (src/main.cpp:5592-6495): the camera rows of Cx, then the tables
        int32_t n_rows = 0;
        if (ba_covariance_camera_rows(p, &n_rows, nullptr, nullptr, nullptr, nullptr)) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
        std::vector<int32_t> par((size_t)n_rows + 1);
        std::vector<double> ce((size_t)n_rows * 6 * d.n_sta + 1), cx((size_t)n_rows * 3 * d.n_pts + 1), cc((size_t)n_rows * n_rows + 1);
        if (ba_covariance_camera_rows(p, nullptr, par.data(), ce.data(), cx.data(), cc.data())) { fprintf(stderr, "error: %s\n", ba_last_error()); return 1; }
        ba_correlations corr;
        // one slot per station in the reference's matrices, overwritten camera by camera: the last camera's figures
        if (ba_correlations_compute(d.n_sta, d.n_pts, d.n_cam - 1, n_rows, par.data(), ve.data(), vx.data(), vi.data(), va.data(), ce.data(),
                                    cx.data(), cc.data(), &corr) ||
            ba_print_correlations(&corr, nullptr)) {
          fprintf(stderr, "error: %s\n", ba_host_last_error());
          return 1;
        }
      }
    }
  }
  std::vector<double> res(2 * (size_t)d.n_obs + 3 * (size_t)d.n_pts + 3 * (size_t)d.n_cam);
  double cost = 0;
  if (ba_evaluate(p, ba_net_eop(net), ba_net_xyz(net), ba_net_iop(net), ba_net_ap(net), 1, &cost, res.data())) {
    fprintf(stderr, "error: %s\n", ba_last_error());
    return 1;
  }
  // what the reference prints and writes after the solve: angles.jck (:4756-4782), the variance-factor, AIC/BIC and
  // quantile lines (:4905-5137), the reprojection errors (:6982-6984), the result files, the QC against truth (:7242)
  ba_report rep;
  std::vector<ba_bin_stats> bin_stats(bins > 0 ? bins : 1);
  // (several ranks: the statistics mix counts of the whole network with those of the shard, so they are left out)
  if (ba_net_write_angles(net, out.c_str()) ||
      (ranks == 1 && (ba_net_report(net, &sm, res.data(), (int64_t)res.size(), bins > 0 ? bins : 0, &rep, bin_stats.data()) ||
                      ba_net_print_report(net, &rep, bin_stats.data(), nullptr))) ||
      ba_net_write_results(net, out.c_str(), res.data())) {
    fprintf(stderr, "error: %s\n", ba_host_last_error());
    return 1;
  }
  if (!truth.empty()) {
    ba_qc qc;
    if (ba_net_qc(net, truth.c_str(), &qc) || ba_net_print_qc(&qc, truth.c_str(), nullptr)) {
      fprintf(stderr, "error: %s\n", ba_host_last_error());
      return 1;
    }
  }
  ba_destroy(p);
  ba_net_free(net);
  return 0;
}
