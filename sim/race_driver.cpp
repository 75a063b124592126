// Race-detection driver of the functional simulator (test infrastructure; see sim/hip/hip_runtime.h "Race detection").
// Built with -fsanitize=thread together with the engine sources: every wave is a ThreadSanitizer fiber, barriers and
// launch boundaries are the only happens-before edges, so a report is an LDS location shared by two waves without a
// barrier, or a global location shared by two blocks of one launch that run on different workers (blocks are dealt
// statically, block b to worker b mod HIPSIM_THREADS, and no block starts before every worker holds the job; run with
// two coprime worker counts to pair every two blocks on different workers).  Runs forward passes of the three networks (random
// weights: only the access pattern matters) through the C ABI, both arithmetic modes, and the stand-alone projection /
// confusion-matrix operators.  Exit status 0 = no report (TSAN_OPTIONS=exitcode=66 otherwise).
//   usage: race_driver [ssv2|dn21|dn53|ops ...]      (default: all four, small shapes)
//          race_driver full                         the three benchmark workloads at full size
//          race_driver threads                      two host threads with a handle each (thread-safety of the library's globals)
//          race_driver selftest                     positive and negative controls of the detector itself
#include "../pclsegmentation_amd/csrc/pclseg_api.hip"      // one translation unit: the engine, then the driver

// ThreadSanitizer calls this for every report (tsan debugging interface)
static int g_reports = 0;
extern "C" void __tsan_on_report(void*) { ++g_reports; }

// ---- controls: what the detector must and must not report
namespace pclseg {     // (the kernels' `extern __shared__ smem_raw` names pclseg::smem_raw, the simulator's LDS)
__global__ void lds_no_barrier(int* out) {      // wave 1 reads what wave 0 wrote, no barrier: a race
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* sm = reinterpret_cast<int*>(smem_raw);
  const int t = threadIdx.x;
  if (t < 64) sm[t] = t;
  if (t >= 64) out[t - 64] = sm[t - 64];
}
__global__ void lds_with_barrier(int* out) {    // the same with the barrier: clean
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* sm = reinterpret_cast<int*>(smem_raw);
  const int t = threadIdx.x;
  if (t < 64) sm[t] = t;
  __syncthreads();
  if (t >= 64) out[t - 64] = sm[t - 64];
}
__global__ void lds_within_a_wave(int* out) {   // lanes of ONE wave exchange through LDS around a wave operation: clean
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* sm = reinterpret_cast<int*>(smem_raw);
  const int t = threadIdx.x;
  sm[t] = t;
  const int u = __shfl_xor(t, 1);               // (every lane has stored once this returns)
  out[blockIdx.x * 64 + t] = sm[t ^ 1] + u;
}
__global__ void global_shared_word(int* out) {  // every block stores to the same word, not atomically: a race
  if (threadIdx.x == 0) out[0] = (int)blockIdx.x;
}
__global__ void global_atomic_word(unsigned* out) {   // the same with an atomic: clean
  if (threadIdx.x == 0) atomicOr(out, 1u << (blockIdx.x & 31));
}
}  // namespace pclseg

static int run_selftest() {
  int* buf = nullptr;
  if (hipMalloc(&buf, 64 * 64 * sizeof(int)) != hipSuccess) return 2;
  memset(buf, 0, 64 * 64 * sizeof(int));
  struct { const char* name; bool racy; int before, after; } row[5] = {
    {"LDS, two waves, no barrier", true, 0, 0}, {"LDS, two waves, barrier", false, 0, 0}, {"LDS, lanes of one wave", false, 0, 0},
    {"global word, 64 blocks, plain stores", true, 0, 0}, {"global word, 64 blocks, atomics", false, 0, 0}};
  row[0].before = g_reports; hipLaunchKernelGGL(pclseg::lds_no_barrier, dim3(1), dim3(128), 256, nullptr, buf); row[0].after = g_reports;
  row[1].before = g_reports; hipLaunchKernelGGL(pclseg::lds_with_barrier, dim3(1), dim3(128), 256, nullptr, buf); row[1].after = g_reports;
  row[2].before = g_reports; hipLaunchKernelGGL(pclseg::lds_within_a_wave, dim3(8), dim3(64), 256, nullptr, buf); row[2].after = g_reports;
  // (ThreadSanitizer is a dynamic detector: it can miss a race when two of its contexts have shared a clock slot, so the
  // racy launch is repeated a few times — one report is what is asked for)
  row[3].before = g_reports;
  for (int rep = 0; rep < 8 && g_reports == row[3].before; ++rep) hipLaunchKernelGGL(pclseg::global_shared_word, dim3(64), dim3(64), 0, nullptr, buf);
  row[3].after = g_reports;

  row[4].before = g_reports; hipLaunchKernelGGL(pclseg::global_atomic_word, dim3(64), dim3(64), 0, nullptr, reinterpret_cast<unsigned*>(buf + 8)); row[4].after = g_reports;
  int bad = 0;
  for (int i = 0; i < 5; ++i) {
    auto& r = row[i];
    const bool reported = r.after > r.before;
    printf("selftest: %-40s %s (%s)\n", r.name, reported ? "REPORTED" : "clean", reported == r.racy ? "as it must be" : "WRONG");
    bad += reported != r.racy;
  }
  (void)hipFree(buf);
  // ---- the stream model (HIPSIM_STREAMS): a consumer on the null stream with and without a dependency on its producer
  {
    hipStream_t side = nullptr;
    hipEvent_t ev = nullptr;
    unsigned char *dev = nullptr, host[2] = {0, 0};
    if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) || hipEventCreateWithFlags(&ev, hipEventDisableTiming) || hipMalloc(&dev, 64)) return 2;
    (void)hipMemset(dev, 1, 64);
    (void)hipMemsetAsync(dev, 2, 64, side);                      // producer on a side stream ...
    (void)hipMemcpy(&host[0], dev, 1, hipMemcpyDeviceToHost);    // ... consumer on the null stream, NO dependency
    (void)hipEventRecord(ev, side);
    (void)hipStreamWaitEvent(nullptr, ev, 0);                    // now the null stream waits for the producer
    (void)hipMemcpy(&host[1], dev, 1, hipMemcpyDeviceToHost);
    const int mode = hipsim_streams_lazy();
    const bool stale_seen = host[0] == 1, fresh_after_wait = host[1] == 2;
    const bool ok = fresh_after_wait && (stale_seen == (mode != 0));
    printf("selftest: %-40s %s (%s)\n", mode ? "stream model: un-joined producer" : "stream model: eager",
           stale_seen ? "consumer ran first, saw stale data" : "consumer saw the producer's data", ok ? "as it must be" : "WRONG");
    bad += !ok;
    (void)hipStreamDestroy(side); (void)hipEventDestroy(ev); (void)hipFree(dev);
  }
  return bad ? 3 : 0;
}

static thread_local uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static float urand() {
  g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
  return (float)((g_rng >> 40) / 16777216.0);
}

#define CHECK(rc, h) do { if ((rc) != PCLSEG_OK) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, pclseg_last_error(h)); exit(2); } } while (0)

static void run_net(const char* label, int arch, int h, int w, int nc, int n, unsigned flags) {
  pclseg_desc d;
  memset(&d, 0, sizeof d);
  d.arch = arch; d.height = h; d.width = w; d.num_class = nc; d.none_index = 0; d.output_stride = 16; d.micro_batch = 2; d.flags = flags;
  for (int i = 0; i < 5; ++i) { d.mean[i] = 0.5 * i; d.std[i] = 1.0 + i; }
  pclseg_handle* hd = nullptr;
  CHECK(pclseg_create(&d, &hd), nullptr);
  const int nw = pclseg_num_weights(hd);
  for (int i = 0; i < nw; ++i) {
    char name[256];
    int64_t shape[4];
    int ndim = 0;
    CHECK(pclseg_weight_info(hd, i, name, sizeof name, shape, &ndim), hd);
    size_t count = 1, fan_in = 1;
    for (int k = 0; k < ndim; ++k) count *= (size_t)shape[k];
    for (int k = 0; k + 1 < ndim; ++k) fan_in *= (size_t)shape[k];
    std::vector<float> v(count);
    const std::string nm(name);
    const bool var = nm.find("variance") != std::string::npos, gamma = nm.find("gamma") != std::string::npos;
    const float scale = ndim >= 3 ? 1.7f / sqrtf((float)fan_in) : 0.1f;
    for (auto& x : v) x = (var || gamma) ? 0.8f + 0.4f * urand() : scale * (2.f * urand() - 1.f);
    CHECK(pclseg_set_weight(hd, name, v.data(), shape, ndim), hd);
  }
  CHECK(pclseg_finalize(hd), hd);
  const size_t px = (size_t)n * h * w;
  std::vector<float> scans(px * 5), logits(px * nc), probs(px * nc);
  for (size_t i = 0; i < px; ++i) {
    const bool valid = urand() < 0.8f;
    for (int c = 0; c < 5; ++c) scans[i * 5 + c] = valid ? 4.f * urand() + (c == 4 ? 0.5f : -2.f) : 0.f;
  }
  std::vector<int32_t> preds(px);
  std::vector<uint8_t> mask(px);
  CHECK(pclseg_forward_raw(hd, scans.data(), n, preds.data(), probs.data(), logits.data(), mask.data(), PCLSEG_MEM_HOST), hd);
  CHECK(pclseg_forward_raw(hd, scans.data(), n, preds.data(), nullptr, nullptr, nullptr, PCLSEG_MEM_HOST), hd);
  long sum = 0;
  for (size_t i = 0; i < px; ++i) sum += preds[i];
  CHECK(pclseg_destroy(hd), nullptr);
  printf("%-28s %dx%d x %d scans, flags %u: done (prediction checksum %ld)\n", label, h, w, n, flags, sum);
  fflush(stdout);
}

static void run_ops() {
  const int h = 16, w = 64, m = 4000, nc = 11;
  std::vector<float> pts((size_t)m * 4), img((size_t)h * w * 5);
  for (auto& x : pts) x = 40.f * urand() - 20.f;
  std::vector<int32_t> idx((size_t)h * w);
  std::vector<uint64_t> scratch((size_t)h * w);
  if (pclseg_op_project(pts.data(), (size_t)m, h, w, 3.0f, -25.0f, 0.0f, img.data(), idx.data(), scratch.data(), nullptr) != PCLSEG_OK) {
    fprintf(stderr, "FAIL project: %s\n", pclseg_last_error(nullptr)); exit(2);
  }
  const size_t count = 20000;
  std::vector<int32_t> lab(count), prd(count);
  for (size_t i = 0; i < count; ++i) { lab[i] = (int)(urand() * nc); prd[i] = (int)(urand() * nc); }
  std::vector<int64_t> cm((size_t)nc * nc, 0);
  if (pclseg_op_confusion_matrix(lab.data(), prd.data(), count, nc, cm.data(), nullptr) != PCLSEG_OK) {
    fprintf(stderr, "FAIL confusion: %s\n", pclseg_last_error(nullptr)); exit(2);
  }
  long long tot = 0;
  for (auto v : cm) tot += v;
  printf("%-28s projection %d points -> %dx%d, confusion matrix %lld / %zu counted: done\n", "operators", m, h, w, tot, count);
}

int main(int argc, char** argv) {
  std::vector<std::string> what;
  for (int i = 1; i < argc; ++i) what.push_back(argv[i]);
  if (argc == 2 && !strcmp(argv[1], "selftest")) return run_selftest();
  auto want = [&](const char* k) { if (what.empty()) return true; for (auto& s : what) if (s == k) return true; return false; };
  if (want("ssv2")) {
    run_net("SqueezeSegV2 f16x3", PCLSEG_ARCH_SQUEEZESEGV2, 64, 256, 20, 3, 0);
    run_net("SqueezeSegV2 f32", PCLSEG_ARCH_SQUEEZESEGV2, 32, 240, 11, 3, PCLSEG_FLAG_EXACT_F32);
  }
  if (want("dn21")) {
    run_net("Darknet-21 f16x3", PCLSEG_ARCH_DARKNET21, 32, 128, 20, 2, 0);
  }
  if (want("dn53")) {
    run_net("Darknet-53 f16x3", PCLSEG_ARCH_DARKNET53, 16, 64, 20, 2, 0);
    run_net("Darknet-53 f32", PCLSEG_ARCH_DARKNET53, 16, 64, 20, 2, PCLSEG_FLAG_EXACT_F32);
  }
  if (want("ops")) run_ops();
  if (want("threads") && !what.empty()) {      // two host threads, a handle each, at the same time: the library's process-wide state
    std::thread t1([] { run_net("host thread 1: SqueezeSegV2", PCLSEG_ARCH_SQUEEZESEGV2, 32, 240, 11, 2, 0); });
    std::thread t2([] { run_net("host thread 2: Darknet-21", PCLSEG_ARCH_DARKNET21, 16, 64, 20, 2, PCLSEG_FLAG_RANGE_FALLBACK); });
    t1.join(); t2.join();
  }
  if (!what.empty() && what[0] == "full") {   // the benchmark workloads at their full size (BASELINE.json configs[1], [4], [2]); slow
    run_net("SqueezeSegV2 f16x3 full", PCLSEG_ARCH_SQUEEZESEGV2, 64, 2048, 20, 4, 0);
    run_net("Darknet-21 f16x3 full", PCLSEG_ARCH_DARKNET21, 32, 1024, 20, 2, 0);
    run_net("Darknet-53 f16x3 full", PCLSEG_ARCH_DARKNET53, 64, 2048, 20, 1, 0);
  }
  printf("race_driver: all passes completed, %d ThreadSanitizer report(s)\n", g_reports);
  return g_reports ? 66 : 0;
}
