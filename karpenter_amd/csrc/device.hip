// device.hip -- the small gfx950 kernels and the HIP implementation of hal.hpp (the solve kernels: dev_part.hip).
#include <hip/hip_runtime.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "hal.hpp"
#include "dev_kernels.hpp"

namespace kg {

// k_type_filter: requirement-set entries x instance types -> T-bit masks (one thread per output word)
__global__ void __launch_bounds__(256) k_type_filter(DevProblem P, PrepassTables X) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = X.E * (uint32_t)P.TW;
  if (i >= total) return;
  const uint32_t e = i / P.TW;
  const int w = (int)(i % P.TW);
  X.e_amask_out[i] = type_filter_word(P, e, w);
}

// k_node_static: requirement sets x existing nodes -> N-bit masks.  One wave covers 64 consecutive
// nodes of one requirement set; the ballot is the output word.
__global__ void __launch_bounds__(256) k_node_static(DevProblem P, PrepassTables X) {
  const uint32_t s = blockIdx.y;
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = n < (uint32_t)P.N && node_static_bit(P, X, s, n);
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && (n >> 6) < (uint32_t)P.NW) X.nlab_out[(size_t)s * P.NW + (n >> 6)] = b;
}

// k_pod_mask: pod variants x instance types -> the prefilter's per-variant type masks (one thread per output word)
__global__ void __launch_bounds__(256) k_pod_mask(DevProblem P, PrepassTables X) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = (uint32_t)P.NV * (uint32_t)P.TW;
  if (i >= total) return;
  X.podmask_out[i] = pod_mask_word(P, i / P.TW, (int)(i % P.TW));
}

// k_pod_ctx: pod variants -> contiguous pod-context images (one thread per 8-byte word; after k_type_filter / k_pod_mask)
__global__ void __launch_bounds__(256) k_pod_ctx(DevProblem P, PrepassTables X) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per = sizeof(PodCtx) / 8;
  if (i >= (uint32_t)P.NV * per || P.TW > KG_TWLDS) return;
  ((uint64_t*)X.pctx_out)[i] = pod_ctx_word(P, i / per, i % per);
}

// k_price_order_truncate: one workgroup per new NodeClaim (solve_core.hpp: InstanceTypes.OrderByPrice / Truncate)
__global__ void __launch_bounds__(256) k_price_order_truncate(DevProblem P, const SimState* sims, TruncArgs X) {
  __shared__ double price[KG_TRUNC_T];
  __shared__ uint16_t nrank[KG_TRUNC_T];
  __shared__ uint32_t nset;
  const SimState S = sims[0];
  const uint32_t c = blockIdx.x, tid = threadIdx.x, T = (uint32_t)P.T;
  if (c >= S.header[0] || T > KG_TRUNC_T) return;
  uint64_t zmask, ctmask;
  claim_offering_masks(P, S, c, &zmask, &ctmask);
  if (tid == 0) nset = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t t = tid; t < T; t += 256) {
    const bool in = (S.c_mask[(size_t)(t >> 6) * P.Cmax + c] >> (t & 63)) & 1;
    price[t] = in ? type_cheapest(X, t, zmask, ctmask) : -1.0;  // (prices are never negative: -1 = not an option of this claim)
    nrank[t] = X.name_rank[t];
    mine += in;
  }
  if (mine) atomicAdd(&nset, mine);
  __syncthreads();
  for (uint32_t t = tid; t < T; t += 256) {
    const double pt = price[t];
    if (pt < 0) continue;
    const uint32_t rt = nrank[t];
    uint32_t rank = 0;
    for (uint32_t u = 0; u < T; ++u) {
      const double pu = price[u];
      rank += (pu >= 0 && price_before(pu, nrank[u], pt, rt)) ? 1u : 0u;
    }
    if (rank < X.max_types) X.rec[(size_t)c * X.rec_words + 4 + rank] = t;
  }
  trunc_pack(P, S, X, c, tid, 256, nset);
}

// k_sim_init: one workgroup per simulation copies the pristine state image into its slot and applies the patches
__global__ void __launch_bounds__(256) k_sim_init(DevProblem P, SimBatch B) {
  sim_init_copy(B, blockIdx.x, threadIdx.x, blockDim.x);
  __syncthreads();
  sim_init_patch(P, B, blockIdx.x, threadIdx.x, blockDim.x);
}
// k_sim_gather: one wave per simulation condenses the outcome into a SimResult (the lanes share the walk over the
// simulation's pods: a multi-node prefix has more than a thousand)
__global__ void __launch_bounds__(64) k_sim_gather(DevProblem P, SimBatch B, const SimState* sims) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  const SimState S = sims[i];
  uint32_t errors = 0, claim0_np = 0;
  sim_gather_scan(B, S, i, lane, 64, &errors, &claim0_np);
  for (int o = 32; o; o >>= 1) errors += __shfl_xor(errors, o);
  claim0_np = __ballot(claim0_np != 0) != 0;
  if (lane == 0) sim_gather_finish(P, B, S, i, errors, claim0_np);
}

}  // namespace kg

// Device memory of kg_solve_noalloc (include/karpgpu.h): the per-solve tables that kg_solve keeps in pooled hipMalloc
// buffers (problem image, solve state, per-NodeClaim rows) live here instead, so the call allocates nothing.  Every GPU the
// library runs on holds its own copy of the array until the process ends; the host only ever uses the per-device address
// hipGetSymbolAddress returned at kg_init (`g_scratch_ptr`), never the symbol itself.
constexpr size_t KG_SCRATCH_BYTES = 32u << 20;
static __device__ __align__(256) unsigned char scratch[KG_SCRATCH_BYTES];

namespace kg {
namespace hal {

static void ck(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

constexpr int KG_MAX_DEVICES = 64;
static std::mutex g_dev_mu;
static void* g_scratch_ptr[KG_MAX_DEVICES];             // per device: address of ITS copy of `scratch`
static std::vector<hipEvent_t> g_events[KG_MAX_DEVICES];  // per device: timing events, created at init and reused
static hipEvent_t g_noalloc_events[KG_MAX_DEVICES][2];   // per device: the pair kg_solve_noalloc's launch groups use (under g_scratch_mu)
static std::mutex g_scratch_mu[KG_MAX_DEVICES];          // per device: one user of the scratch array at a time
constexpr int KG_EVENT_POOL = 32;                         // events per device: 4 per consolidation round / 2 per launch group in flight
static thread_local bool t_noalloc = false;
// The stream of the context whose entry point the calling thread is in (set with the device by CtxLock): every copy, fill,
// launch and event of the call goes to it, so calls on different contexts overlap on the GPU and none touches the null stream.
static thread_local hipStream_t t_stream = nullptr;

std::string init(int device, int* device_out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return "no HIP device available (libkarpgpu has no CPU fallback)";
  if (device >= 0) {
    e = hipSetDevice(device);
    if (e != hipSuccess) return std::string("hipSetDevice: ") + hipGetErrorString(e);
  }
  int dev = 0;
  e = hipGetDevice(&dev);
  if (e != hipSuccess) return std::string("hipGetDevice: ") + hipGetErrorString(e);
  if (dev < 0 || dev >= KG_MAX_DEVICES) return "device index out of range";
  *device_out = dev;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));  // (with `dev` current: each GPU has its own copy at its own address)
    if (e != hipSuccess || !p) return std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
    g_scratch_ptr[dev] = p;
    for (int i = 0; i < KG_EVENT_POOL + 2; ++i) {
      hipEvent_t ev;
      e = hipEventCreate(&ev);
      if (e != hipSuccess) return std::string("hipEventCreate: ") + hipGetErrorString(e);
      if (i < 2) g_noalloc_events[dev][i] = ev; else g_events[dev].push_back(ev);
    }
  }
  return "";
}
void use_device(int device, void* stream) {
  (void)hipSetDevice(device);
  t_stream = (hipStream_t)stream;
}
int device_count() {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
void* stream_create() {
  hipStream_t s = nullptr;
  ck(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
  return s;
}
void stream_destroy(void* s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
void* scratch_ptr(size_t* bytes) {
  int dev = 0;
  ck(hipGetDevice(&dev), "hipGetDevice");
  std::lock_guard<std::mutex> lk(g_dev_mu);
  if (dev < 0 || dev >= KG_MAX_DEVICES || !g_scratch_ptr[dev]) throw std::runtime_error("scratch array not set up on this device (kg_init)");
  *bytes = KG_SCRATCH_BYTES;
  return g_scratch_ptr[dev];
}
std::mutex& scratch_mutex() {
  int dev = 0;
  ck(hipGetDevice(&dev), "hipGetDevice");
  if (dev < 0 || dev >= KG_MAX_DEVICES) throw std::runtime_error("device index out of range");
  return g_scratch_mu[dev];
}
void set_noalloc(bool on) { t_noalloc = on; }
void* dmalloc(size_t bytes) {
  void* p = nullptr;
  ck(hipMalloc(&p, bytes ? bytes : 8), "hipMalloc");
  return p;
}
void dfree(void* p) { if (p) (void)hipFree(p); }
void* host_pinned(size_t bytes) {
  void* p = nullptr;
  ck(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault), "hipHostMalloc");
  return p;
}
void host_pinned_free(void* p) { if (p) (void)hipHostFree(p); }
// (copies complete before they return: the callers hand over pageable host buffers they reuse at once)
void h2d(void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  ck(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, t_stream), "hipMemcpy H2D");
  ck(hipStreamSynchronize(t_stream), "hipMemcpy H2D");
}
void h2d_async(void* dst, const void* src_pinned, size_t bytes) {
  if (bytes) ck(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, t_stream), "hipMemcpy H2D (page-locked)");
}
void sync() { ck(hipStreamSynchronize(t_stream), "hipStreamSynchronize"); }
void d2h(void* dst, const void* src, size_t bytes) {
  if (!bytes) return;
  ck(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, t_stream), "hipMemcpy D2H");
  ck(hipStreamSynchronize(t_stream), "hipMemcpy D2H");
}
void d2h_rows(void* dst, const void* src, size_t width, size_t spitch, size_t rows) {
  if (!width || !rows) return;
  ck(hipMemcpy2DAsync(dst, width, src, spitch, width, rows, hipMemcpyDeviceToHost, t_stream), "hipMemcpy2D D2H");
  ck(hipStreamSynchronize(t_stream), "hipMemcpy2D D2H");
}
void dzero(void* dst, size_t bytes) { if (bytes) ck(hipMemsetAsync(dst, 0, bytes, t_stream), "hipMemset"); }
void dfill(void* dst, int byte, size_t bytes) { if (bytes) ck(hipMemsetAsync(dst, byte, bytes, t_stream), "hipMemset"); }

// roctx ranges (rocprofv3 --marker-trace): every launch group and, from the host side, every entry point's phases
void range_push(const char* name) { roctxRangePushA(name); }
void range_pop() { roctxRangePop(); }

// HIP-event bracket of one launch group; the events come from the device's pool (created at kg_init) and go back to it.
// Doubles as the group's roctx range.
// N events of the current device for one launch group + its roctx range; both go back when the scope ends, also when a
// launch or a wait throws in between.
template <int N>
struct EventScope {
  hipEvent_t ev[N];
  int dev = 0, taken = 0;
  bool reserved = false;
  explicit EventScope(const char* name) {
    roctxRangePushA(name);
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KG_MAX_DEVICES) { roctxRangePop(); throw std::runtime_error("hipGetDevice"); }
    if (t_noalloc && N <= 2) {  // kg_solve_noalloc: the pair set aside at kg_init (its caller holds the device's scratch mutex)
      for (int i = 0; i < N; ++i) ev[i] = g_noalloc_events[dev][i];
      reserved = true; taken = N;
      return;
    }
    std::unique_lock<std::mutex> lk(g_dev_mu);
    while (taken < N) {
      if (!g_events[dev].empty()) { ev[taken++] = g_events[dev].back(); g_events[dev].pop_back(); continue; }
      lk.unlock();  // (more launch groups in flight on this device than pooled events: many concurrent contexts)
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) { release(); throw std::runtime_error("hipEventCreate"); }
      ev[taken++] = e;
      lk.lock();
    }
  }
  void release() {
    if (!reserved) {
      std::lock_guard<std::mutex> lk(g_dev_mu);
      for (int i = 0; i < taken; ++i) g_events[dev].push_back(ev[i]);
    }
    taken = 0;
    roctxRangePop();
  }
  ~EventScope() { release(); }
  EventScope(const EventScope&) = delete;
  EventScope& operator=(const EventScope&) = delete;
};
struct Timer {
  EventScope<2> es;
  explicit Timer(const char* name) : es(name) { ck(hipEventRecord(es.ev[0], t_stream), "record"); }
  float stop() {
    ck(hipEventRecord(es.ev[1], t_stream), "record");
    ck(hipEventSynchronize(es.ev[1]), "kernel execution");
    float ms = 0;
    ck(hipEventElapsedTime(&ms, es.ev[0], es.ev[1]), "elapsed");
    return ms;
  }
};

void launch_prepass(const DevProblem& P, const PrepassTables& X, float* ms) {
  Timer t("kg prepass (k_type_filter, k_node_static, k_pod_mask, k_pod_ctx)");
  const uint32_t total = X.E * (uint32_t)P.TW;
  if (total) hipLaunchKernelGGL(k_type_filter, dim3((total + 255) / 256), dim3(256), 0, t_stream, P, X);
  if (P.N > 0 && P.S > 0) hipLaunchKernelGGL(k_node_static, dim3((P.N + 255) / 256, P.S), dim3(256), 0, t_stream, P, X);
  const uint32_t pm = (uint32_t)P.NV * (uint32_t)P.TW;
  if (pm) hipLaunchKernelGGL(k_pod_mask, dim3((pm + 255) / 256), dim3(256), 0, t_stream, P, X);
  const uint32_t pcw = (uint32_t)P.NV * (uint32_t)(sizeof(PodCtx) / 8);
  if (pcw && X.pctx_out && P.TW <= KG_TWLDS) hipLaunchKernelGGL(k_pod_ctx, dim3((pcw + 255) / 256), dim3(256), 0, t_stream, P, X);
  ck(hipGetLastError(), "prepass launch");
  *ms = t.stop();
}

void launch_solve(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int block, float* ms, bool batch) {
  Timer t("kg solve (k_solve / k_simulate)");
  if (batch) {
    enqueue_k_simulate(P, sims, nsims, nqueue, block, false, t_stream);
    ck(hipGetLastError(), "simulate launch");
    *ms = t.stop();
    return;
  }
  if (P.stats & 2) launch_probes(P, sims, t_stream);  // (debug_probes.hip)
  enqueue_k_solve(P, sims, nsims, nqueue, block, (P.stats & (2 | 8)) != 0, t_stream);  // (debug timers: 256 threads only)
  ck(hipGetLastError(), "solve launch");
  *ms = t.stop();
}
void launch_solve_wave(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, float* ms) {
  Timer t("kg solve (k_solve_wave)");
  enqueue_k_solve_wave(P, sims, nsims, nqueue, (P.stats & 8) != 0, t_stream);
  ck(hipGetLastError(), "solve (wave) launch");
  *ms = t.stop();
}
void launch_solve_spec(const DevProblem& P, const SimState* sims, int nsims, const uint32_t* nqueue, int waves, float* ms) {
  Timer t("kg solve (k_solve_spec)");
  enqueue_k_solve_spec(P, sims, nsims, nqueue, waves, (P.stats & 8) != 0, t_stream);  // (debug: per-wave cycle accounting)
  ck(hipGetLastError(), "solve (speculative) launch");
  *ms = t.stop();
}
void launch_truncate(const DevProblem& P, const SimState* sims, const TruncArgs& X, int max_claims) {
  // (no timer of its own: it runs behind the solve kernel on the same stream, and the decode's first copy waits for both)
  if (max_claims > 0) hipLaunchKernelGGL(k_price_order_truncate, dim3(max_claims), dim3(256), 0, t_stream, P, sims, X);
  ck(hipGetLastError(), "truncate launch");
}
void launch_sim_init(const DevProblem& P, const SimBatch& B, float* ms) {
  Timer t("kg k_sim_init");
  if (B.nsims) hipLaunchKernelGGL(k_sim_init, dim3(B.nsims), dim3(256), 0, t_stream, P, B);
  ck(hipGetLastError(), "sim init launch");
  *ms = t.stop();
}
void launch_sim_gather(const DevProblem& P, const SimBatch& B, const SimState* sims, float* ms) {
  Timer t("kg k_sim_gather");
  if (B.nsims) hipLaunchKernelGGL(k_sim_gather, dim3(B.nsims), dim3(64), 0, t_stream, P, B, sims);
  ck(hipGetLastError(), "sim gather launch");
  *ms = t.stop();
}
// k_sim_init -> k_simulate -> k_sim_gather of one batch back to back: four events, ONE host synchronisation (the three
// launch groups above each wait for their own kernel)
void launch_sim_round(const DevProblem& P, const SimBatch& B, const SimState* sims, int block, float ms[3], void* res_host, const void* res_dev, size_t res_bytes) {
  ms[0] = ms[1] = ms[2] = 0;
  if (!B.nsims) return;
  EventScope<4> es("kg consolidation round (k_sim_init, k_simulate, k_sim_gather)");
  hipEvent_t* ev = es.ev;
  ck(hipEventRecord(ev[0], t_stream), "record");
  hipLaunchKernelGGL(k_sim_init, dim3(B.nsims), dim3(256), 0, t_stream, P, B);
  ck(hipEventRecord(ev[1], t_stream), "record");
  enqueue_k_simulate(P, sims, B.nsims, B.nqueue, block, (P.stats & 8) != 0, t_stream);  // (debug: per-wave cycle accounting)
  ck(hipEventRecord(ev[2], t_stream), "record");
  hipLaunchKernelGGL(k_sim_gather, dim3(B.nsims), dim3(64), 0, t_stream, P, B, sims);
  ck(hipEventRecord(ev[3], t_stream), "record");
  ck(hipGetLastError(), "consolidation round launch");
  if (res_host && res_bytes) {  // (the outcomes ride behind the gather: one wait for kernels and copy)
    ck(hipMemcpyAsync(res_host, res_dev, res_bytes, hipMemcpyDeviceToHost, t_stream), "hipMemcpy D2H");
    ck(hipStreamSynchronize(t_stream), "kernel execution");
  } else ck(hipEventSynchronize(ev[3]), "kernel execution");
  for (int i = 0; i < 3; ++i) ck(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]), "elapsed");
}
const char* backend() { return "hip-gfx950"; }

}  // namespace hal
}  // namespace kg
