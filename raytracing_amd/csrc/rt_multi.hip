// rt_multi.hip — several GPUs behind one handle, for a single-process caller (a JVM).
//
// The reference fans a frame out over one thread per image row and joins them on a latch
// (core.clj:109-128). Here the frame is fanned out over the GPUs of a node: one host thread per
// device renders that device's shard (its share of the 8x8 tiles, or its slice of the samples)
// through rt_render_device, the per-device FP32 accumulators are combined onto the first device
// with one RCCL reduce(sum) over xGMI, and the first device packs the pixels. A sample's radiance
// depends only on (seed, pixel, sample), so the split never changes what is computed.
//
// RCCL is loaded on first use (dlopen), and only when the devices are distinct and more than one:
// a single GPU needs no collective, and a list that names one device several times (a rehearsal
// of the fan-out on a one-GPU machine) is summed by a kernel on that device, in list order.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/rtmi.h"

namespace {

// ---- the few RCCL entry points used, bound at run time ------------------------------------
typedef void* nccl_comm;
struct Rccl {
  void* so = nullptr;
  int (*CommInitAll)(nccl_comm*, int, const int*) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
  // One handle for the process, opened on first need and kept: RCCL registers its device code with the HIP runtime
  // when it is loaded, and unloading it under a live runtime is not something every ROCm release survives, so it is
  // never closed; every rt_multi handle and the selftest share it, so repeated use opens nothing more.
  static Rccl& shared() { static Rccl r; return r; }
  static std::mutex& lock() { static std::mutex m; return m; }
  bool load() {
    std::lock_guard<std::mutex> guard(lock());
    if (so) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (so) break;
    }
    if (!so) { why = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "not found"); return false; }
    auto sym = [&](const char* n) { void* p = dlsym(so, n); if (!p) why = std::string("librccl.so lacks ") + n; return p; };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    Reduce = reinterpret_cast<decltype(Reduce)>(sym("ncclReduce"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !Reduce || !GroupStart || !GroupEnd || !GetErrorString) { dlclose(so); so = nullptr; return false; }
    return true;
  }
};
constexpr int kNcclFloat = 7, kNcclSum = 0;  // ncclFloat32, ncclSum of nccl.h

thread_local std::string g_multi_err;
int mfail(int code, const char* fmt, ...) {
  char buf[640];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_multi_err = buf;
  return code;
}

__global__ void k_accumulate(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

}  // namespace

struct rt_multi {
  std::vector<int> devices;
  std::vector<rt_ctx*> ctx;
  std::vector<hipStream_t> stream;
  std::vector<float*> d_sum;      // per device: the W*H*3 accumulator of its shard
  std::vector<size_t> sum_bytes;
  uint32_t* d_argb = nullptr;     // on the first device
  size_t argb_bytes = 0;
  bool distinct = true;           // no device named twice
  std::vector<nccl_comm> comm;
};
struct rt_multi_scene {
  rt_multi* m = nullptr;
  std::vector<rt_scene*> scene;
};

extern "C" {

const char* rt_multi_last_error(void) { return g_multi_err.c_str(); }

void rt_multi_shutdown(rt_multi* m) {
  if (!m) return;
  for (size_t i = 0; i < m->ctx.size(); ++i) {
    if (!m->ctx[i]) continue;  // (never bound: its device id may not even exist)
    hipSetDevice(m->devices[i]);
    if (i < m->comm.size() && m->comm[i]) Rccl::shared().CommDestroy(m->comm[i]);
    if (i < m->d_sum.size() && m->d_sum[i]) hipFree(m->d_sum[i]);
    if (i < m->stream.size() && m->stream[i]) hipStreamDestroy(m->stream[i]);
    if (m->ctx[i]) rt_shutdown(m->ctx[i]);
  }
  if (m->d_argb) { hipSetDevice(m->devices[0]); hipFree(m->d_argb); }
  delete m;
}

int rt_multi_init(const int32_t* device_ids, int32_t n_devices, rt_multi** out) {
  if (!out) return mfail(RT_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64) return mfail(RT_ERR_ARG, "need 1..64 device ids");
  try {
    rt_multi* m = new rt_multi();
    std::set<int> seen;
    for (int i = 0; i < n_devices; ++i) {
      m->devices.push_back(device_ids[i]);
      if (!seen.insert(device_ids[i]).second) m->distinct = false;
    }
    m->ctx.assign((size_t)n_devices, nullptr);
    m->stream.assign((size_t)n_devices, nullptr);
    m->d_sum.assign((size_t)n_devices, nullptr);
    m->sum_bytes.assign((size_t)n_devices, 0);
    for (int i = 0; i < n_devices; ++i) {
      int rc = rt_init(device_ids[i], &m->ctx[(size_t)i]);
      if (rc) { g_multi_err = std::string("device ") + std::to_string(device_ids[i]) + ": " + rt_last_error(); rt_multi_shutdown(m); return rc; }
      if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreate(&m->stream[(size_t)i]) != hipSuccess) {
        rt_multi_shutdown(m);
        return mfail(RT_ERR_HIP, "cannot create a stream on device %d", device_ids[i]);
      }
    }
    if (n_devices > 1 && m->distinct) {
      if (!Rccl::shared().load()) { std::string why = Rccl::shared().why; rt_multi_shutdown(m); return mfail(RT_ERR_HIP, "%s", why.c_str()); }
      m->comm.assign((size_t)n_devices, nullptr);
      int e = Rccl::shared().CommInitAll(m->comm.data(), n_devices, m->devices.data());
      if (e != 0) {
        std::string why = Rccl::shared().GetErrorString(e);
        m->comm.clear();
        rt_multi_shutdown(m);
        return mfail(RT_ERR_HIP, "ncclCommInitAll failed: %s", why.c_str());
      }
    }
    *out = m;
    return RT_OK;
  } catch (const std::bad_alloc&) {
    return mfail(RT_ERR_OOM, "out of host memory in rt_multi_init");
  } catch (...) {
    return mfail(RT_ERR_ARG, "rt_multi_init: unknown failure");
  }
}

// Loads RCCL and runs the very calls rt_multi_render makes (ncclCommInitAll, a grouped in-place
// ncclReduce(float, sum, root 0), ncclCommDestroy) on ONE device: a reduce over one rank must
// return its input. What a one-GPU machine can check of the collective path.
int rt_multi_selftest_rccl(int32_t device_id) {
  Rccl& R = Rccl::shared();
  if (!R.load()) return mfail(RT_ERR_HIP, "%s", R.why.c_str());
  if (hipSetDevice(device_id) != hipSuccess) return mfail(RT_ERR_HIP, "hipSetDevice(%d) failed", device_id);
  nccl_comm comm = nullptr;
  int dev = device_id;
  int e = R.CommInitAll(&comm, 1, &dev);
  if (e != 0) return mfail(RT_ERR_HIP, "ncclCommInitAll failed: %s", R.GetErrorString(e));
  const size_t n = 3 * 1000 + 1;
  float* d = nullptr;
  hipStream_t stream = nullptr;
  std::vector<float> h(n), back(n, -1.f);
  for (size_t i = 0; i < n; ++i) h[i] = 0.25f * (float)i - 7.f;
  int rc = RT_OK;
  if (hipMalloc(&d, n * 4) != hipSuccess || hipStreamCreate(&stream) != hipSuccess) rc = mfail(RT_ERR_HIP, "selftest allocation failed");
  if (!rc && hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) rc = mfail(RT_ERR_HIP, "selftest upload failed");
  if (!rc) {
    e = R.GroupStart();
    if (e == 0) e = R.Reduce(d, d, n, kNcclFloat, kNcclSum, 0, comm, stream);
    const int e2 = R.GroupEnd();
    if (e != 0 || e2 != 0) rc = mfail(RT_ERR_HIP, "ncclReduce failed: %s", R.GetErrorString(e ? e : e2));
  }
  if (!rc && (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(back.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess))
    rc = mfail(RT_ERR_HIP, "selftest readback failed");
  if (!rc && std::memcmp(h.data(), back.data(), n * 4) != 0) rc = mfail(RT_ERR_HIP, "a float sum-reduce over one rank did not return its input");
  if (stream) hipStreamDestroy(stream);
  if (d) hipFree(d);
  R.CommDestroy(comm);
  return rc;
}

int rt_multi_device_count(const rt_multi* m) { return m ? (int)m->devices.size() : 0; }

void rt_multi_scene_destroy(rt_multi_scene* s) {
  if (!s) return;
  for (rt_scene* sc : s->scene) rt_scene_destroy(sc);
  delete s;
}

int rt_multi_scene_create(rt_multi* m, const rt_scene_desc* desc, rt_multi_scene** out) {
  if (!m || !out) return mfail(RT_ERR_ARG, "handle or out is NULL");
  *out = nullptr;
  try {
    rt_multi_scene* s = new rt_multi_scene();
    s->m = m;
    s->scene.assign(m->ctx.size(), nullptr);
    for (size_t i = 0; i < m->ctx.size(); ++i) {  // the world is replicated: it is KiB to MiB
      int rc = rt_scene_create(m->ctx[i], desc, &s->scene[i]);
      if (rc) { g_multi_err = rt_last_error(); rt_multi_scene_destroy(s); return rc; }
    }
    *out = s;
    return RT_OK;
  } catch (const std::bad_alloc&) {
    return mfail(RT_ERR_OOM, "out of host memory in rt_multi_scene_create");
  }
}

int rt_multi_render(rt_multi* m, const rt_multi_scene* s, const rt_camera* cam, const rt_params* params, int32_t split,
                    float* rgb_sum, uint32_t* argb, rt_stats* stats) {
  if (!m || !s || s->m != m || !cam || !params) return mfail(RT_ERR_ARG, "handle, scene, camera or params is NULL");
  if (split != RT_SPLIT_TILES && split != RT_SPLIT_SAMPLES) return mfail(RT_ERR_ARG, "unknown split %d", split);
  if (params->shard_count > 1 || params->shard_index != 0)
    return mfail(RT_ERR_ARG, "rt_multi_render shards the frame itself: pass shard_index 0, shard_count 0 or 1");
  if (params->width <= 0 || params->height <= 0 || params->width > 65535 || params->height > 65535 ||
      (long long)params->width * params->height > (1ll << 28) || params->spp <= 0)
    return mfail(RT_ERR_ARG, "image size %dx%d or spp %d out of range", params->width, params->height, params->spp);
  try {
    const int n = (int)m->devices.size();
    const size_t npix = (size_t)params->width * params->height, bytes = npix * 3 * sizeof(float);
    for (int i = 0; i < n; ++i) {
      if (m->sum_bytes[(size_t)i] >= bytes) continue;
      if (hipSetDevice(m->devices[(size_t)i]) != hipSuccess) return mfail(RT_ERR_HIP, "hipSetDevice(%d) failed", m->devices[(size_t)i]);
      if (m->d_sum[(size_t)i]) hipFree(m->d_sum[(size_t)i]);
      m->d_sum[(size_t)i] = nullptr;
      m->sum_bytes[(size_t)i] = 0;
      if (hipMalloc(&m->d_sum[(size_t)i], bytes) != hipSuccess) return mfail(RT_ERR_OOM, "no room for the accumulator on device %d", m->devices[(size_t)i]);
      m->sum_bytes[(size_t)i] = bytes;
    }
    if (argb && m->argb_bytes < npix * 4) {
      hipSetDevice(m->devices[0]);
      if (m->d_argb) hipFree(m->d_argb);
      m->d_argb = nullptr;
      m->argb_bytes = 0;
      if (hipMalloc(&m->d_argb, npix * 4) != hipSuccess) return mfail(RT_ERR_OOM, "no room for the packed frame on device %d", m->devices[0]);
      m->argb_bytes = npix * 4;
    }
    // ---- fan out: one host thread per device (the reference's Worker threads, core.clj:44-48,109-127)
    std::vector<rt_params> shard((size_t)n, *params);
    std::vector<rt_stats> st((size_t)n);
    std::vector<int> rc((size_t)n, RT_OK);
    std::vector<std::string> why((size_t)n);
    for (int i = 0; i < n; ++i) {
      rt_params& p = shard[(size_t)i];
      if (split == RT_SPLIT_TILES) {
        p.shard_index = i;
        p.shard_count = n;
      } else {
        const int lo = (int)((long long)params->spp * i / n), hi = (int)((long long)params->spp * (i + 1) / n);
        p.spp = hi - lo;
        p.sample_offset = params->sample_offset + lo;
      }
    }
    auto work = [&](int i) {
      const rt_params& p = shard[(size_t)i];
      if (p.spp <= 0) {  // fewer samples than devices: this one contributes nothing
        hipSetDevice(m->devices[(size_t)i]);
        hipMemsetAsync(m->d_sum[(size_t)i], 0, bytes, m->stream[(size_t)i]);
        std::memset(&st[(size_t)i], 0, sizeof(rt_stats));
        return;
      }
      rc[(size_t)i] = rt_render_device(m->ctx[(size_t)i], s->scene[(size_t)i], cam, &p, m->d_sum[(size_t)i], nullptr, &st[(size_t)i], m->stream[(size_t)i]);
      if (rc[(size_t)i]) why[(size_t)i] = rt_last_error();
    };
    if (n == 1 || !m->distinct) {
      // (contexts that share a device share nothing else, but their renders run one after the other)
      for (int i = 0; i < n; ++i) work(i);
    } else {
      // one host thread per device; a thread that cannot be started (std::system_error) must not unwind past the
      // ones already running: they are joined first, and the device does its share on this thread instead
      std::vector<std::thread> threads;
      threads.reserve((size_t)n);
      std::vector<int> inline_work;
      for (int i = 0; i < n; ++i) {
        try {
          threads.emplace_back(work, i);
        } catch (const std::system_error&) {
          inline_work.push_back(i);
        }
      }
      for (int i : inline_work) work(i);
      for (auto& t : threads) t.join();
    }
    for (int i = 0; i < n; ++i)
      if (rc[(size_t)i]) return mfail(rc[(size_t)i], "device %d: %s", m->devices[(size_t)i], why[(size_t)i].c_str());
    // ---- combine the accumulators on the first device
    if (n > 1 && m->distinct) {
      int e = Rccl::shared().GroupStart();
      for (int i = 0; i < n && e == 0; ++i) {
        hipSetDevice(m->devices[(size_t)i]);
        e = Rccl::shared().Reduce(m->d_sum[(size_t)i], m->d_sum[(size_t)i], npix * 3, kNcclFloat, kNcclSum, 0, m->comm[(size_t)i], m->stream[(size_t)i]);
      }
      const int e2 = Rccl::shared().GroupEnd();
      if (e != 0 || e2 != 0) return mfail(RT_ERR_HIP, "ncclReduce failed: %s", Rccl::shared().GetErrorString(e ? e : e2));
      for (int i = 0; i < n; ++i) {
        hipSetDevice(m->devices[(size_t)i]);
        if (hipStreamSynchronize(m->stream[(size_t)i]) != hipSuccess) return mfail(RT_ERR_HIP, "device %d failed during the reduce", m->devices[(size_t)i]);
      }
    } else if (n > 1) {
      hipSetDevice(m->devices[0]);
      for (int i = 0; i < n; ++i) hipStreamSynchronize(m->stream[(size_t)i]);
      for (int i = 1; i < n; ++i)
        hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((npix * 3 + 255) / 256)), dim3(256), 0, m->stream[0], m->d_sum[0], m->d_sum[(size_t)i], npix * 3);
      if (hipStreamSynchronize(m->stream[0]) != hipSuccess) return mfail(RT_ERR_HIP, "the sum of the accumulators failed");
    }
    // ---- the pixel post-process on the first device (core.clj:106-108,118,121-122), then home
    hipSetDevice(m->devices[0]);
    if (argb) {
      int r = rt_tonemap_device(m->ctx[0], m->d_sum[0], (int32_t)npix, params->spp, m->d_argb, m->stream[0]);
      if (r) return mfail(r, "%s", rt_last_error());
      if (hipMemcpyAsync(argb, m->d_argb, npix * 4, hipMemcpyDeviceToHost, m->stream[0]) != hipSuccess) return mfail(RT_ERR_HIP, "copy of the packed frame failed");
    }
    if (rgb_sum && hipMemcpyAsync(rgb_sum, m->d_sum[0], bytes, hipMemcpyDeviceToHost, m->stream[0]) != hipSuccess)
      return mfail(RT_ERR_HIP, "copy of the sums failed");
    if (hipStreamSynchronize(m->stream[0]) != hipSuccess) return mfail(RT_ERR_HIP, "the first device failed after the reduce");
    if (stats) {
      std::memset(stats, 0, sizeof *stats);
      for (int i = 0; i < n; ++i) {
        stats->samples += st[(size_t)i].samples;
        stats->rays += st[(size_t)i].rays;
        for (int e = 0; e < RT_EV_COUNT; ++e) stats->events[e] += st[(size_t)i].events[e];
        stats->kernel_ms = std::max(stats->kernel_ms, st[(size_t)i].kernel_ms);  // the devices run side by side
        stats->shader_mhz = std::max(stats->shader_mhz, st[(size_t)i].shader_mhz);
      }
    }
    return RT_OK;
  } catch (const std::bad_alloc&) {
    return mfail(RT_ERR_OOM, "out of host memory in rt_multi_render");
  } catch (const std::exception& e) {
    return mfail(RT_ERR_HIP, "rt_multi_render: %s", e.what());
  }
}

}  // extern "C"
