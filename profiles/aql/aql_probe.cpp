// What does a kernel launch cost when the AQL packet is written by hand (HSA user-mode queue) instead of through hipLaunchKernel?
// Host time per dispatch, kernel-to-kernel gap on one queue for the three fence scopes, doorbell -> first wave, last wave -> host.
// build: g++ -O2 -I/opt/rocm/include aql_probe.cpp -L/opt/rocm/lib -lhsa-runtime64 ; run: ./aql_probe spin_kernel.hsaco
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(s_, &m); fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m); exit(1); } } while (0)

static hsa_agent_t g_gpu, g_cpu;
static hsa_amd_memory_pool_t g_kernarg_pool, g_dev_pool;
static bool g_have_gpu = false, g_have_cpu = false, g_have_ka = false, g_have_dev = false;

static hsa_status_t on_agent(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t flags = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_ka) { g_kernarg_pool = p; g_have_ka = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t flags = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev) { g_dev_pool = p; g_have_dev = true; }
    return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object; uint32_t kernarg_size, group_size, private_size; };

static uint64_t now_ticks() { uint64_t t; hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP, &t); return t; }

struct Args { unsigned long long ticks; unsigned int* out; };

static void submit(hsa_queue_t* q, const Kernel& k, void* kernarg, uint32_t grid, int acquire, int release, bool barrier, hsa_signal_t done) {
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
    while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
    hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
    p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p->workgroup_size_x = 64; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->grid_size_x = grid; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = k.private_size; p->group_segment_size = k.group_size;
    p->kernel_object = k.object; p->kernarg_address = kernarg; p->reserved2 = 0; p->completion_signal = done;
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                       (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    __atomic_store_n((uint16_t*)&p->header, header, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "spin_kernel.hsaco";
    CK(hsa_init());
    CK(hsa_iterate_agents(on_agent, nullptr));
    if (!g_have_gpu || !g_have_cpu) { fprintf(stderr, "no agents\n"); return 1; }
    hsa_amd_agent_iterate_memory_pools(g_cpu, on_cpu_pool, nullptr);
    hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr);
    if (!g_have_ka || !g_have_dev) { fprintf(stderr, "no pools\n"); return 1; }
    uint64_t freq = 0; hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq);
    char name[64] = {0}; hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name);
    printf("agent %s, timestamp frequency %.1f MHz\n", name, freq / 1e6);

    FILE* f = fopen(path, "rb"); if (!f) { perror(path); return 1; }
    fseek(f, 0, SEEK_END); const long len = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blob(len); if (fread(blob.data(), 1, len, f) != (size_t)len) return 1; fclose(f);
    hsa_code_object_reader_t reader; CK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    hsa_executable_t exe; CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    CK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    CK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym; CK(hsa_executable_get_symbol_by_name(exe, "k_spin.kd", &g_gpu, &sym));
    Kernel k{};
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_size));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_size));
    printf("kernel object %#llx kernarg %u B group %u B private %u B\n", (unsigned long long)k.object, k.kernarg_size, k.group_size, k.private_size);

    hsa_queue_t* q; CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_MULTI, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    CK(hsa_amd_profiling_set_profiler_enabled(q, 1));
    unsigned int* out; CK(hsa_amd_memory_pool_allocate(g_dev_pool, 4096, 0, (void**)&out));
    const int RING = 4096;
    // kernel arguments: in the host's kernarg pool (every wave then reads them over PCIe), or — argv[2] == "dev" — in device memory
    // that the CPU writes through the PCIe BAR (what HIP does on this GPU: HIP_FORCE_DEV_KERNARG)
    const bool dev_kernarg = argc > 2 && !strcmp(argv[2], "dev");
    char* ka;
    if (dev_kernarg) {
        CK(hsa_amd_memory_pool_allocate(g_dev_pool, (size_t)RING * 64 + 4096, 0, (void**)&ka));
        CK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, ka));
    } else {
        CK(hsa_amd_memory_pool_allocate(g_kernarg_pool, (size_t)RING * 64 + 4096, 0, (void**)&ka));
        CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ka));
    }
    printf("kernel arguments in %s memory\n", dev_kernarg ? "device (written through the BAR)" : "host");
    std::vector<hsa_signal_t> sig(64);
    for (auto& s : sig) CK(hsa_signal_create(1, 0, nullptr, &s));
    auto arg_at = [&](int i, unsigned long long ticks) {
        Args* a = (Args*)(ka + (size_t)(i % RING) * 64); a->ticks = ticks; a->out = out;
        if (dev_kernarg) { __builtin_ia32_sfence(); (void)*(volatile unsigned int**)&a->out; }      // write-combined BAR writes out, then a read back behind them
        return (void*)a; };
    const hsa_signal_t none = {0};

    // warm-up
    hsa_signal_store_relaxed(sig[0], 1);
    submit(q, k, arg_at(0, 0), 1024 * 64, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM, true, sig[0]);
    hsa_signal_wait_scacquire(sig[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);

    // 0. grid sizes, before and after 60 ms of load on every CU
    for (int phase = 0; phase < 2; phase++) {
        if (phase == 1) {
            hsa_signal_store_relaxed(sig[0], 1);
            for (int i = 0; i < 1200; i++) submit(q, k, arg_at(i, 5000), 8192 * 64, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, true, i == 1199 ? sig[0] : none);
            hsa_signal_wait_scacquire(sig[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
        }
        const uint32_t grids[4] = {1, 256, 1024, 4096};
        for (int g = 0; g < 4; g++)
            for (unsigned long long ticks = 0; ticks <= 400; ticks += 400) {
                const int n = 200;
                hsa_signal_store_relaxed(sig[0], 1);
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; i++) submit(q, k, arg_at(i, ticks), grids[g] * 64, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, true, i == n - 1 ? sig[0] : none);
                hsa_signal_wait_scacquire(sig[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
                auto t2 = std::chrono::steady_clock::now();
                printf("%s: %4u workgroups, kernel spins %llu ticks: %.3f us per launch\n", phase ? "after load" : "cold", grids[g], ticks,
                       std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
            }
    }
    // 1. host cost per dispatch (empty kernels, 1024 workgroups of one wave, as k_chain<1>)
    for (int rep = 0; rep < 3; rep++) {
        const int n = 2000;
        hsa_signal_store_relaxed(sig[0], 1);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) submit(q, k, arg_at(i, 0), 1024 * 64, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, true, i == n - 1 ? sig[0] : none);
        auto t1 = std::chrono::steady_clock::now();
        hsa_signal_wait_scacquire(sig[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
        auto t2 = std::chrono::steady_clock::now();
        printf("host: %.3f us per dispatch to enqueue %d empty launches; all done after %.3f us per launch\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / n, n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    }
    // 1b. the same with 4 us kernels, still no profiling: host clock over 64 launches
    for (int rep = 0; rep < 3; rep++)
        for (int s = 0; s < 3; s++) {
            const int scopes_[3] = {HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE};
            const int n = 64;
            hsa_signal_store_relaxed(sig[0], 1);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) submit(q, k, arg_at(i, 400), 1024 * 64, scopes_[s], scopes_[s], true, i == n - 1 ? sig[0] : none);
            hsa_signal_wait_scacquire(sig[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
            auto t2 = std::chrono::steady_clock::now();
            printf("no profiling, fence scope %d, 4 us kernels on one queue: %.3f us per launch (host clock, %d launches)\n", scopes_[s],
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / n, n);
        }
    // 2. kernel-to-kernel gap on one queue (barrier bit), 4 us kernels, per fence scope; 3. doorbell -> start, end -> host
    const int scopes[3] = {HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE};
    const char* scope_name[3] = {"system", "agent", "none"};
    for (int rep = 0; rep < 2; rep++)
        for (int s = 0; s < 3; s++)
            for (int barrier = 1; barrier >= 0; barrier--) {
                const int n = 32;
                for (int i = 0; i < n; i++) hsa_signal_store_relaxed(sig[i], 1);
                const uint64_t h0 = now_ticks();
                for (int i = 0; i < n; i++) submit(q, k, arg_at(i, 400), 1024 * 64, scopes[s], scopes[s], barrier != 0, sig[i]);
                hsa_signal_wait_scacquire(sig[n - 1], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
                const uint64_t h1 = now_ticks();
                for (int i = 0; i < n; i++) hsa_signal_wait_scacquire(sig[i], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
                std::vector<hsa_amd_profiling_dispatch_time_t> t(n);
                for (int i = 0; i < n; i++) CK(hsa_amd_profiling_get_dispatch_time(g_gpu, sig[i], &t[i]));
                double gap = 0, dur = 0;
                for (int i = 1; i < n; i++) gap += (double)((int64_t)(t[i].start - t[i - 1].end));
                for (int i = 0; i < n; i++) dur += (double)(t[i].end - t[i].start);
                const double us = 1e6 / (double)freq;
                printf("fences %-6s barrier %d: kernel %.2f us, gap end->next start %.2f us, period %.2f us; first enqueue -> first start %.2f us; last end -> host saw it %.2f us\n",
                       scope_name[s], barrier, dur / n * us, gap / (n - 1) * us, (double)(t[n - 1].end - t[0].start) / n * us,
                       (double)((int64_t)(t[0].start - h0)) * us, (double)((int64_t)(h1 - t[n - 1].end)) * us);
            }
    // 4. does the doorbell -> first wave latency depend on how long the queue has been idle?
    for (int idle_us : {0, 20, 100, 500, 2000}) {
        double lat = 0;
        const int n = 8;
        for (int r = 0; r < n; r++) {
            auto w0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count() < idle_us) {}
            hsa_signal_store_relaxed(sig[0], 1);
            const uint64_t h0 = now_ticks();
            submit(q, k, arg_at(r, 100), 1024 * 64, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, true, sig[0]);
            hsa_signal_wait_scacquire(sig[0], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
            hsa_amd_profiling_dispatch_time_t t; CK(hsa_amd_profiling_get_dispatch_time(g_gpu, sig[0], &t));
            lat += (double)((int64_t)(t.start - h0)) * 1e6 / (double)freq;
        }
        printf("queue idle for %4d us: enqueue -> first wave %.2f us\n", idle_us, lat / n);
    }
    unsigned int count = 0; CK(hsa_memory_copy(&count, out, 4));
    printf("kernels that ran: %u\n", count);
    hsa_queue_destroy(q);
    hsa_shut_down();
    return 0;
}
