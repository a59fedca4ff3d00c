// r03: WHY does the headline kernel's time follow which 2 GB blocks its two outputs live in?  
// Standalone (no torch): separately hipMalloc'ed 2.048 GB buffers, timed per buffer and per combination with
//   * a plain streaming kernel (pure write / pure read / K2's 1 read : 2 writes mix), and
//   * K2 itself through the C-ABI of libvbmp_hip.so (dlopen),
// then the same with odd virtual skews inside one pool, with deliberately fragmented VRAM, and with VMM-mapped memory.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/place_probe.hip -o tools/exp/place_probe -ldl
//   tools/exp/place_probe [section ...]     sections: perbuf mix k2 skew frag vmm pmc   (default: all but pmc)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <string>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static const size_t NB = (size_t)2048 * 1000 * 1000;  // bytes per stream = 1e6 x 16 x 16 doubles
static const size_t NV = NB / 16;

template <int NR, int NW>
__global__ __launch_bounds__(256) void k_mix(const f4* __restrict__ r0, f4* __restrict__ w0, f4* __restrict__ w1, size_t n, float* sink) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    f4 v = {1, 2, 3, 4};
    if (NR >= 1) v = r0[i];
    if (NW >= 1) w0[i] = v;
    if (NW >= 2) w1[i] = v * 2.0f;
    if (NW == 0) acc += v;
  }
  if (NW == 0 && acc.x == 123.456f) *sink = acc.y;
}

// K2's memory skeleton: a wave reads CH x 1 KB (16 B per lane per instruction), waits, idles `d1` x ~1.7 us, writes the tile to w0,
// idles `d2`, writes it to w1.  One tile per wave, WPB waves per block; `extra LDS` limits the blocks per CU.
template <int CH, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_skel(const f4* __restrict__ r, f4* __restrict__ w0, f4* __restrict__ w1, size_t ntiles, int d1, int d2) {
  extern __shared__ float dyn[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t tile = (size_t)blockIdx.x * WPB + wave;
  if (tile >= ntiles) return;
  const size_t base = tile * (64 * CH) + lane;
  f4 v[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) v[i] = r[base + 64 * i];
  float x = v[0].x;
#pragma unroll
  for (int i = 1; i < CH; ++i) x += v[i].x;   // waits for every load
  for (int k = 0; k < d1; ++k) { __builtin_amdgcn_s_sleep(64); asm volatile("" : "+v"(x)); }
  v[0].x = x;
#pragma unroll
  for (int i = 0; i < CH; ++i) w0[base + 64 * i] = v[i];
  for (int k = 0; k < d2; ++k) { __builtin_amdgcn_s_sleep(64); asm volatile("" : "+v"(x)); }
  v[0].y = x;
#pragma unroll
  for (int i = 0; i < CH; ++i) w1[base + 64 * i] = v[i] * 2.0f;
  if (d1 < 0) dyn[threadIdx.x] = x;
}

// SPD inputs for K2: SExx = 40 I + small symmetric part, SEx small, N = 32
__global__ void k_init_spd(double* SExx, double* SEx, double* N, size_t B) {
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= B) return;
  for (int i = 0; i < 16; ++i) {
    for (int j = 0; j < 16; ++j) {
      double off = 0.01 * (double)(((m + 3 * (i + j) + i * j) % 17)) - 0.08;
      SExx[m * 256 + i * 16 + j] = (i == j) ? 40.0 + (double)(m % 5) : off;
    }
    SEx[m * 16 + i] = 0.1 * (double)((m + i) % 7) - 0.3;
  }
  N[m] = 32.0;
}

static hipEvent_t E0, E1;
static float* g_sink;

template <typename F>
static float med(F&& f, int reps = 5, int warm = 1) {
  std::vector<float> ts;
  for (int it = 0; it < reps + warm; ++it) {
    CK(hipEventRecord(E0));
    f();
    CK(hipEventRecord(E1));
    CK(hipEventSynchronize(E1));
    float ms; CK(hipEventElapsedTime(&ms, E0, E1));
    if (it >= warm) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}
static const int FULLGRID = (int)((NV + 255) / 256);
static float t_write(void* w, int blocks = FULLGRID) { return med([&] { hipLaunchKernelGGL((k_mix<0, 1>), dim3(blocks), dim3(256), 0, 0, (const f4*)w, (f4*)w, (f4*)w, NV, g_sink); }); }
static float t_read(void* r, int blocks = FULLGRID) { return med([&] { hipLaunchKernelGGL((k_mix<1, 0>), dim3(blocks), dim3(256), 0, 0, (const f4*)r, (f4*)r, (f4*)r, NV, g_sink); }); }
static float t_mix(void* r, void* w0, void* w1, int blocks = FULLGRID) { return med([&] { hipLaunchKernelGGL((k_mix<1, 2>), dim3(blocks), dim3(256), 0, 0, (const f4*)r, (f4*)w0, (f4*)w1, NV, g_sink); }); }

// ---- K2 through the C-ABI
typedef int (*niw_fn)(const double*, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t,
                      const double*, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t, const double*, int64_t,
                      const double*, int64_t, double, double*, double*, double*, double*, double*, double*, int64_t, int, int, int*, void*);
static niw_fn g_niw;
static void (*g_set_cap)(int);
static double *g_SEx, *g_N, *g_lam0, *g_mu0, *g_i0, *g_nu0, *g_lam, *g_mu, *g_nu, *g_logdet;
static const int64_t BB = 1000000;
static float t_k2(void* SExx, void* invU, void* U, int reps = 5) {
  return med([&] {
    int rc = g_niw((const double*)SExx, 256, g_SEx, 16, g_N, 1, g_lam0, 0, g_mu0, 0, g_i0, 0, g_nu0, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0,
                   1.0, g_lam, g_mu, (double*)invU, g_nu, (double*)U, g_logdet, BB, 16, 0, nullptr, nullptr);
    if (rc != 0) { printf("K2 rc %d\n", rc); exit(1); }
  }, reps);
}
template <int CH, int WPB>
static float t_skel(void* r, void* w0, void* w1, int d1, int d2, int lds_kb) {
  const size_t ntiles = NV / (64 * CH);
  const int blocks = (int)((ntiles + WPB - 1) / WPB);
  if (lds_kb > 64) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_skel<CH, WPB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
  return med([&] { hipLaunchKernelGGL((k_skel<CH, WPB>), dim3(blocks), dim3(64 * WPB), (size_t)lds_kb * 1024, 0, (const f4*)r, (f4*)w0, (f4*)w1, ntiles, d1, d2); });
}
static double pct(double bytes, float ms) { return bytes / ms / 1e6 / 80.0; }

static void* vmm_alloc(size_t bytes, size_t chunk, std::vector<hipMemGenericAllocationHandle_t>& hs) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (chunk < gran) chunk = gran;
  chunk = (chunk + gran - 1) / gran * gran;
  size_t total = (bytes + chunk - 1) / chunk * chunk;
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
    hs.push_back(h);
  }
  hipMemAccessDesc ad = {};
  ad.location.type = hipMemLocationTypeDevice;
  ad.location.id = 0;
  ad.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &ad, 1));
  printf("  vmm: %zu bytes as chunks of %zu (granularity %zu) at %p\n", total, chunk, gran, va);
  return va;
}

int main(int argc, char** argv) {
  std::vector<std::string> sec;
  for (int i = 1; i < argc; ++i) sec.push_back(argv[i]);
  auto want = [&](const char* s) { return sec.empty() ? (strcmp(s, "pmc") != 0 && strcmp(s, "frag") != 0 && strcmp(s, "vmm") != 0 && strcmp(s, "skew") != 0) : std::find(sec.begin(), sec.end(), s) != sec.end(); };
  CK(hipEventCreate(&E0)); CK(hipEventCreate(&E1));
  CK(hipMalloc(&g_sink, 4));
  size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
  printf("device memory: free %.1f GiB of %.1f GiB\n", fr / 1073741824.0, tot / 1073741824.0);

  void* so = dlopen("pyvbmp_amd/libvbmp_hip.so", RTLD_NOW);
  if (!so) { printf("dlopen: %s\n", dlerror()); return 1; }
  g_niw = (niw_fn)dlsym(so, "vbmp_niw_ss_update_f64");
  g_set_cap = (void (*)(int))dlsym(so, "vbmp_debug_set_blocks_per_cu");
  if (!g_niw) { printf("no vbmp_niw_ss_update_f64\n"); return 1; }

  const int NBUF = 8;
  void* buf[NBUF];
  for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&buf[i], NB)); printf("buf %d at %p\n", i, buf[i]); }
  CK(hipMalloc(&g_SEx, BB * 16 * 8)); CK(hipMalloc(&g_N, BB * 8)); CK(hipMalloc(&g_lam, BB * 8)); CK(hipMalloc(&g_mu, BB * 16 * 8));
  CK(hipMalloc(&g_nu, BB * 8)); CK(hipMalloc(&g_logdet, BB * 8));
  CK(hipMalloc(&g_lam0, 8)); CK(hipMalloc(&g_mu0, 16 * 8)); CK(hipMalloc(&g_i0, 256 * 8)); CK(hipMalloc(&g_nu0, 8));
  {
    double one = 1.0, nu0 = 18.0, z[16] = {0}, I[256] = {0};
    for (int i = 0; i < 16; ++i) I[i * 17] = 1.0;
    CK(hipMemcpy(g_lam0, &one, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(g_nu0, &nu0, 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_mu0, z, 128, hipMemcpyHostToDevice)); CK(hipMemcpy(g_i0, I, 2048, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(k_init_spd, dim3((BB + 255) / 256), dim3(256), 0, 0, (double*)buf[0], g_SEx, g_N, (size_t)BB);
  CK(hipDeviceSynchronize());
  for (int i = 1; i < NBUF; ++i) CK(hipMemset(buf[i], 0, NB));
  CK(hipDeviceSynchronize());

  if (want("perbuf")) {
    printf("\n== perbuf: pure write / pure read per buffer (one block per 256 chunks), %% of 8 TB/s\n");
    for (int rep = 0; rep < 2; ++rep)
      for (int i = 1; i < NBUF; ++i) {
        float tw = t_write(buf[i]), tr = t_read(buf[i]);
        printf("  buf %d: write %.4f ms (%.1f %%)  read %.4f ms (%.1f %%)\n", i, tw, pct(NB, tw), tr, pct(NB, tr));
      }
  }
  if (want("mix")) {
    printf("\n== mix: 1 read (buf 0) : 2 writes (row = w0 buffer, column = w1 buffer), ms; full grid\n");
    for (int a = 1; a < NBUF; ++a) {
      printf("  ");
      for (int b = 1; b < NBUF; ++b) printf("%s ", a == b ? "  -  " : (std::to_string(t_mix(buf[0], buf[a], buf[b])).substr(0, 5)).c_str());
      printf("\n");
    }
    printf("   (%.3f ms = 77 %% of 8 TB/s)\n", 3.0 * NB / 0.77 / 8e9);
  }
  if (want("k2")) {
    printf("\n== k2: K2 via the C-ABI, SExx = buf 0 (row = invU buffer, column = U buffer), ms  [70 %% = %.3f ms]\n", 6432e6 / 0.70 / 8e9);
    for (int a = 1; a < NBUF; ++a) {
      printf("  ");
      for (int b = 1; b < NBUF; ++b) printf("%s ", a == b ? "  -  " : (std::to_string(t_k2(buf[0], buf[a], buf[b], 3)).substr(0, 5)).c_str());
      printf("\n");
    }
    if (g_set_cap) {
      printf("  grid cap A/B within fixed placements (blocks per CU: 0 = one tile per wave):\n");
      int pairs[4][2] = {{1, 2}, {3, 4}, {5, 6}, {6, 7}};
      for (auto& pr : pairs) {
        printf("   invU=buf%d U=buf%d:", pr[0], pr[1]);
        for (int cap : {1, 2, 3, 4, 6, 8, 16, 32, 64, 128, 256, 100000}) { g_set_cap(cap); printf("  cap %d: %.4f", cap, t_k2(buf[0], buf[pr[0]], buf[pr[1]], 7)); }
        g_set_cap(0);
        printf("\n");
      }
    }
  }
  if (want("skew")) {
    printf("\n== skew: three streams inside ONE 3 x 2.25 GiB pool, odd skews of the two outputs (bytes), mix / K2 ms\n");
    void* pool; const size_t span = (size_t)2304 << 20;
    CK(hipMalloc(&pool, 3 * span + (64 << 20)));
    hipLaunchKernelGGL(k_init_spd, dim3((BB + 255) / 256), dim3(256), 0, 0, (double*)pool, g_SEx, g_N, (size_t)BB);
    CK(hipDeviceSynchronize());
    size_t sk[] = {0, 256, 4096, 65536, (1 << 20) + 4096, (2 << 20) + 256 * 37, (16 << 20) + 4096 * 5 + 768, (33 << 20) + 4096 * 129};
    for (size_t s1 : sk)
      for (size_t s2 : {(size_t)0, s1 / 2 / 256 * 256, s1}) {
        char* p = (char*)pool;
        float tm = t_mix(p, p + span + s1, p + 2 * span + s2), tk = t_k2(p, p + span + s1, p + 2 * span + s2, 3);
        printf("  skew invU +%zu, U +%zu: mix %.4f  K2 %.4f\n", s1, s2, tm, tk);
      }
    CK(hipFree(pool));
  }
  if (want("frag")) {
    printf("\n== frag: 2.048 GB buffers carved out of deliberately fragmented free memory\n");
    // fill most of the free memory with small blocks, free every other one, allocate from the holes
    for (size_t piece : {(size_t)64 << 10, (size_t)2 << 20}) {
      const size_t target = (size_t)12 << 30;
      std::vector<void*> ps;
      for (size_t got = 0; got < target; got += piece) { void* q; if (hipMalloc(&q, piece) != hipSuccess) break; ps.push_back(q); }
      for (size_t i = 0; i < ps.size(); i += 2) { CK(hipFree(ps[i])); ps[i] = nullptr; }
      void *fa, *fb;
      CK(hipMalloc(&fa, NB)); CK(hipMalloc(&fb, NB));
      CK(hipMemset(fa, 0, NB)); CK(hipMemset(fb, 0, NB));
      float tw = t_write(fa), tw2 = t_write(fb), tm = t_mix(buf[0], fa, fb), tk = t_k2(buf[0], fa, fb, 5);
      printf("  holes of %zu KiB (%zu pieces kept): write %.4f / %.4f ms, mix %.4f, K2 %.4f   [at %p %p]\n", piece >> 10, ps.size() / 2, tw, tw2, tm, tk, fa, fb);
      CK(hipFree(fa)); CK(hipFree(fb));
      for (void* q : ps) if (q) CK(hipFree(q));
    }
  }
  if (want("vmm")) {
    printf("\n== vmm: outputs as VMM mappings (hipMemCreate chunks)\n");
    for (size_t chunk : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30, (size_t)0}) {
      std::vector<hipMemGenericAllocationHandle_t> hs;
      size_t ck = chunk ? chunk : 2 * ((NB + ((size_t)2 << 20) - 1) / ((size_t)2 << 20)) * ((size_t)2 << 20);
      void* va = vmm_alloc(2 * NB + ((size_t)4 << 20), ck, hs);
      char* p = (char*)va;
      char* p2 = p + ((NB + ((size_t)2 << 20) - 1) / ((size_t)2 << 20)) * ((size_t)2 << 20);
      CK(hipMemset(p, 0, NB)); CK(hipMemset(p2, 0, NB));
      float tw = t_write(p), tw2 = t_write(p2), tm = t_mix(buf[0], p, p2), tk = t_k2(buf[0], p, p2, 5);
      printf("  chunk %zu MiB: write %.4f / %.4f ms, mix %.4f, K2 %.4f\n", ck >> 20, tw, tw2, tm, tk);
      // leave mapped (process ends soon); unmapping order is not what is measured
    }
  }
  if (want("skel")) {
    printf("\n== skel: K2's memory skeleton (read CH KB per wave, idle d1, write, idle d2, write; one tile per wave), ms; mix at 1 KB per wave = %.4f\n", t_mix(buf[0], buf[1], buf[2]));
    for (int lds_kb : {0, 37, 50}) {
      printf("  extra LDS %d KB per block (blocks/CU: %s)\n", lds_kb, lds_kb == 0 ? "8" : lds_kb == 37 ? "4" : "3");
      for (int d : {0, 1, 2, 4}) {
        printf("   idle %d+%d:  4 waves/block: CH1 %.4f  CH2 %.4f  CH4 %.4f  CH8 %.4f   | 1 wave/block: CH8 %.4f  | 2 waves/block: CH8 %.4f\n", d, d,
               t_skel<1, 4>(buf[0], buf[1], buf[2], d, d, lds_kb), t_skel<2, 4>(buf[0], buf[1], buf[2], d, d, lds_kb),
               t_skel<4, 4>(buf[0], buf[1], buf[2], d, d, lds_kb), t_skel<8, 4>(buf[0], buf[1], buf[2], d, d, lds_kb),
               t_skel<8, 1>(buf[0], buf[1], buf[2], d, d, lds_kb / 4), t_skel<8, 2>(buf[0], buf[1], buf[2], d, d, lds_kb / 2));
      }
    }
  }
  if (want("libs")) {
    // A/B builds of the library (tools/exp/build_variant.sh): VBMP_PROBE_LIBS=name=path[:flags],...  interleaved, fixed placement
    const char* env = getenv("VBMP_PROBE_LIBS");
    if (env) {
      struct V { std::string name; niw_fn fn; void (*cap)(int); void (*flags)(int); int fl; int capv; std::vector<float> ts; };
      std::vector<V> vs;
      std::string e(env);
      size_t pos = 0;
      while (pos < e.size()) {
        size_t c = e.find(',', pos); if (c == std::string::npos) c = e.size();
        std::string item = e.substr(pos, c - pos); pos = c + 1;
        size_t eq = item.find('='); std::string name = item.substr(0, eq), rest = item.substr(eq + 1);
        int fl = 0, capv = 0;
        size_t col = rest.find(':');
        if (col != std::string::npos) { sscanf(rest.c_str() + col + 1, "%i:%i", &fl, &capv); rest = rest.substr(0, col); }
        void* h = dlopen(rest.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("dlopen %s: %s\n", rest.c_str(), dlerror()); continue; }
        vs.push_back({name, (niw_fn)dlsym(h, "vbmp_niw_ss_update_f64"), (void (*)(int))dlsym(h, "vbmp_debug_set_blocks_per_cu"), (void (*)(int))dlsym(h, "vbmp_debug_set_flags"), fl, capv, {}});
      }
      printf("\n== libs: interleaved A/B of library builds, invU = buf 1, U = buf 2 and invU = buf 5, U = buf 6 (median of 8 rounds x 3)\n");
      int pairs[2][2] = {{1, 2}, {5, 6}};
      for (auto& pr : pairs) {
        for (auto& v : vs) v.ts.clear();
        for (int rnd = 0; rnd < 8; ++rnd)
          for (auto& v : vs) {
            g_niw = v.fn; v.flags(v.fl); v.cap(v.capv);
            v.ts.push_back(t_k2(buf[0], buf[pr[0]], buf[pr[1]], 3));
            v.flags(0); v.cap(0);
          }
        for (auto& v : vs) { std::sort(v.ts.begin(), v.ts.end()); printf("  pair (%d,%d) %-28s median %.4f  min %.4f  max %.4f\n", pr[0], pr[1], v.name.c_str(), v.ts[v.ts.size() / 2], v.ts[0], v.ts.back()); }
      }
      g_niw = (niw_fn)dlsym(so, "vbmp_niw_ss_update_f64");
    }
  }
  if (want("pmc")) {
    // a fixed dispatch sequence for rocprofv3 --pmc (two launches each, timed as well): the analysis matches by order
    auto two = [&](const char* name, auto&& f) { float t = med(f, 2, 0); printf("pmc %-34s %.4f ms\n", name, t); };
    auto k2 = [&](int cap, int a, int b) {
      char nm[64]; snprintf(nm, sizeof nm, "K2 cap %d invU=buf%d U=buf%d", cap, a, b);
      g_set_cap(cap);
      two(nm, [&] { g_niw((const double*)buf[0], 256, g_SEx, 16, g_N, 1, g_lam0, 0, g_mu0, 0, g_i0, 0, g_nu0, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0,
                          1.0, g_lam, g_mu, (double*)buf[a], g_nu, (double*)buf[b], g_logdet, BB, 16, 0, nullptr, nullptr); });
      g_set_cap(0);
    };
    two("mix (1,2)", [&] { hipLaunchKernelGGL((k_mix<1, 2>), dim3(FULLGRID), dim3(256), 0, 0, (const f4*)buf[0], (f4*)buf[1], (f4*)buf[2], NV, g_sink); });
    two("mix (5,6)", [&] { hipLaunchKernelGGL((k_mix<1, 2>), dim3(FULLGRID), dim3(256), 0, 0, (const f4*)buf[0], (f4*)buf[5], (f4*)buf[6], NV, g_sink); });
    two("skel CH1 (1,2)", [&] { hipLaunchKernelGGL((k_skel<1, 4>), dim3((unsigned)(NV / 64 / 4)), dim3(256), 0, 0, (const f4*)buf[0], (f4*)buf[1], (f4*)buf[2], NV / 64, 0, 0); });
    two("skel CH8 (1,2)", [&] { hipLaunchKernelGGL((k_skel<8, 4>), dim3((unsigned)(NV / 512 / 4)), dim3(256), 0, 0, (const f4*)buf[0], (f4*)buf[1], (f4*)buf[2], NV / 512, 0, 0); });
    for (int cap : {3, 32, 128, 100000}) k2(cap, 1, 2);
    for (int cap : {3, 32, 128, 100000}) k2(cap, 5, 6);
    for (int cap : {32, 100000}) k2(cap, 6, 7);
  }
  return 0;
}
