// Direct dispatch of the chained rollout launches: AQL packets written by this library into HSA user-mode queues of its own,
// instead of hipLaunchKernel on HIP streams (host side only; included by tetris_hip.hip).
//
// Why: a chained launch takes the GPU 4.0 us (one player) and hipLaunchKernel costs its calling thread 2.4-4.2 us, depending on
// the process — long calls needed one enqueue thread per stream to stay ahead of the GPU, and a 20-launch call spends 3-4 us in
// the runtime before its first packet is even visible to the GPU (profiles/r03/host_pace.txt, timing_20_launches.txt).  Written
// by hand a dispatch costs the host 0.2-0.5 us: a 64-byte packet, the kernel arguments, a doorbell (profiles/aql/aql_probe.cpp,
// profiles/r03/aql_probe.txt).  What the runtime did and this file has to do itself:
//  * the code object: the gfx950 image of tetris_hip.hip is taken out of THIS library's own fat binary (the file the dynamic
//    loader mapped; clang offload bundle, uncompressed) and loaded a second time through the HSA loader, which yields the kernel
//    descriptors' addresses.  Same machine code as the HIP path runs; its constant tables are initialised by the loader.
//  * kernel arguments live in DEVICE memory that the host writes through the PCIe BAR (in host memory every one of a launch's
//    1024 waves fetches them over PCIe: 15 us per launch, measured); write-combined stores, one fence and one read back behind
//    them before the doorbell rings.
//  * the hidden arguments of code object v5 behind the explicit ones (block counts, group sizes, grid dims).
//  * ordering: every packet has the barrier bit (a queue's launches run one after the other, as on a stream — chain_fits counts on
//    at most `depth` launches in flight); the first packet of a queue acquires at system scope, everything else is agent scope
//    (rollout_direct in tetris_hip.hip says why that is enough for these kernels).
//  * flow control: every `wgroup`-th packet of a queue carries a completion signal; at most 2 * wgroup + 1 packets per queue are
//    outstanding (the margin of the RNG tables is sized for that, as for the stream path's gate).
// The queues belong to the DEVICE, not to a batch (chained calls of a device's batches exclude each other anyway): three more
// hardware queues per process and GPU, however many batches there are.
// Only calls of at least `direct_min` launches (default 16) go this way.  A queue that has been idle for ~100 us takes 10 us from
// doorbell to first wave instead of 5 (profiles/r03/aql_probe.txt), as a stream's does; in calls as short as the driver's 20
// launches the two paths are level for one player (the queues' times spread less: they do not feel the host's launch cost) and
// the queues 5-8 % ahead for two (profiles/r03/direct_dispatch.txt, (6)); from a few hundred launches on they are 1.5-2 % faster
// and need no helper threads.  A call of a handful of launches is not worth three system-scope acquires.
// Anything that fails while setting this up switches it off for the batch (the stream path remains); TETRIS_DIRECT=0 in the
// environment or tetris_set_direct_dispatch(b, 0) do the same by hand.
#pragma once
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <dlfcn.h>
#include <atomic>
#include <immintrin.h>

namespace aql {

struct Kernel { uint64_t object = 0; uint32_t kernarg = 0, group = 0, priv = 0; bool ok = false; };

struct Queues {                        // one set per device: chained calls of a device's batches exclude each other (chain_acquire)
    hsa_queue_t* q[CHAIN_STREAMS] = {};
    char* kernarg[CHAIN_STREAMS] = {};           // [SLOTS][slot_bytes] each, device memory, host-visible
    uint32_t slot_bytes = 0;
    std::atomic<uint64_t> issued[CHAIN_STREAMS]; // packets ever written to queue k (kernarg slot = issued % SLOTS)
    hsa_signal_t gate[CHAIN_STREAMS][2] = {};    // flow control
    hsa_signal_t done[CHAIN_STREAMS] = {};       // last packet of a call on queue k
    hsa_signal_t first{};                        // first packet of a call (its start time)
    bool ok = false;
    // XCD-affine launches (k_chain_affine): the XCD block 0 of a launch on queue k lands on, measured when the queues are made
    // (blocks are dealt round-robin over the eight XCDs from there); affine_ok: every queue showed that pattern, twice
    uint32_t xcd_base[CHAIN_STREAMS] = {};
    bool affine_ok = false, calibrated = false;
};
struct Device {                        // one per HIP device of the process, made on first use, never torn down
    int hip_device = -1;
    bool tried = false, ok = false;
    hsa_agent_t gpu{}, cpu{};
    hsa_amd_memory_pool_t dev_pool{};
    uint64_t ts_freq = 0;
    std::vector<hsa_executable_t> exes;
    Kernel chain1, chain1_affine, duo, duo_affine, blocker, xcc_probe;
    std::vector<std::vector<char>> images;   // the code objects' bytes: the loader (and a profiler's code-object tracking) keep reading them
    Queues qs;
    std::string why;                   // why it is not ok
};

#ifndef TE_AQL_SLOTS
#define TE_AQL_SLOTS 4096
#endif
constexpr int SLOTS = TE_AQL_SLOTS;    // kernel-argument slots per queue (> 2 * wgroup + 1)
constexpr uint32_t QUEUE_PACKETS = 1024;

static std::mutex g_mutex;
static std::vector<Device*> g_devices;

struct FindCtx { Device* d; uint32_t want_bdf; uint32_t want_domain; bool found_gpu = false, found_cpu = false; };

static hsa_status_t on_agent(hsa_agent_t a, void* p) {
    FindCtx* c = (FindCtx*)p;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !c->found_cpu) { c->d->cpu = a; c->found_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU && !c->found_gpu) {
        uint32_t bdf = 0, domain = 0;
        (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
        if ((bdf & 0xFFFFu) == c->want_bdf && domain == c->want_domain) { c->d->gpu = a; c->found_gpu = true; }
    }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void* out) {
    Device* d = (Device*)out;
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool alloc = false;
    if (hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && d->dev_pool.handle == 0) d->dev_pool = p;
    return HSA_STATUS_SUCCESS;
}
struct SymCtx { Device* d; };
static hsa_status_t on_symbol(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t s, void* p) {
    Device* d = ((SymCtx*)p)->d;
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
    uint32_t len = 0;
    (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len);
    std::string name(len, '\0');
    (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME, &name[0]);
    Kernel* k = nullptr;
    // k_chain<1>(te::KArgs) and k_duo<M_ROLLOUT = 6, true, false>(te::KArgs), Itanium-mangled, with the descriptor suffix
    if (name.rfind("_Z7k_chainILi1EEvN2te5KArgsE", 0) == 0) k = &d->chain1;
    else if (name.rfind("_Z5k_duoILi6ELb1ELb0EEvN2te5KArgsE", 0) == 0) k = &d->duo;
    else if (name.rfind("_Z9k_blockerPKjy", 0) == 0) k = &d->blocker;            // (test aid: tetris_debug_stall)
    else if (name.rfind("_Z14k_chain_affineILi1EEvN2te5KArgsE", 0) == 0) k = &d->chain1_affine;
    else if (name.rfind("tetris_k_xcc_probe", 0) == 0) k = &d->xcc_probe;
    else if (name.rfind("_Z12k_duo_affineN2te5KArgsE", 0) == 0) k = &d->duo_affine;
    if (!k) return HSA_STATUS_SUCCESS;
    bool ok = hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k->object) == HSA_STATUS_SUCCESS;
    ok = ok && hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k->kernarg) == HSA_STATUS_SUCCESS;
    ok = ok && hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k->group) == HSA_STATUS_SUCCESS;
    ok = ok && hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k->priv) == HSA_STATUS_SUCCESS;
    k->ok = ok && k->object != 0;
    return HSA_STATUS_SUCCESS;
}

// the gfx950 code objects inside this shared library's fat binaries (one clang offload bundle per translation unit)
static bool own_code_objects(std::vector<std::vector<char>>& out, std::string& why) {
    Dl_info info;
    if (!dladdr((const void*)&own_code_objects, &info) || !info.dli_fname) { why = "dladdr failed"; return false; }
    FILE* f = fopen(info.dli_fname, "rb");
    if (!f) { why = std::string("cannot read ") + info.dli_fname; return false; }
    std::vector<char> file;
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (len <= 0) { fclose(f); why = "empty library file"; return false; }
    file.resize((size_t)len);
    const size_t got = fread(file.data(), 1, file.size(), f);
    fclose(f);
    if (got != file.size()) { why = "short read of the library file"; return false; }
    static const char magic[] = "__CLANG_OFFLOAD_BUNDLE__";
    const size_t mlen = sizeof magic - 1;
    for (size_t at = 0; at + mlen + 8 <= file.size(); ) {
        const void* hit = memmem(file.data() + at, file.size() - at, magic, mlen);
        if (!hit) break;
        const size_t base = (size_t)((const char*)hit - file.data());
        at = base + mlen;
        uint64_t n = 0;
        memcpy(&n, file.data() + base + mlen, 8);
        size_t o = base + mlen + 8;
        if (n == 0 || n > 64) continue;
        for (uint64_t e = 0; e < n; e++) {
            if (o + 24 > file.size()) break;
            uint64_t off, size, idlen;
            memcpy(&off, file.data() + o, 8); memcpy(&size, file.data() + o + 8, 8); memcpy(&idlen, file.data() + o + 16, 8);
            o += 24;
            if (idlen > 256 || o + idlen > file.size()) break;
            const std::string id(file.data() + o, (size_t)idlen);
            o += (size_t)idlen;
            if (id.find("amdgcn-amd-amdhsa") != std::string::npos && id.find("gfx950") != std::string::npos && size > 0 &&
                base + off + size <= file.size())
                out.emplace_back(file.data() + base + off, file.data() + base + off + size);
        }
    }
    if (out.empty()) { why = "no gfx950 code object found in the library's fat binary"; return false; }
    return true;
}

static Device* device_for(int hip_device) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (Device* d : g_devices) if (d->hip_device == hip_device) return d;
    Device* d = new Device();
    d->hip_device = hip_device;
    d->tried = true;
    g_devices.push_back(d);
    auto bad = [&](const std::string& w) { d->why = w; d->ok = false; return d; };
    {   // A profiler that intercepts HSA queues (rocprofv3 preloads its tool library) stands between a doorbell and the hardware: every
        // packet is copied and rewritten on the way.  One-player runs were traced correctly that way, the two-player bench crashed inside
        // the tool (profiles/r03/direct_dispatch.txt) — with a tool library in the process the launches stay on the streams.
        const char* tools[] = {getenv("ROCP_TOOL_LIBRARIES"), getenv("HSA_TOOLS_LIB"), getenv("LD_PRELOAD")};
        for (const char* t : tools)
            if (t && (strstr(t, "rocprofiler") || strstr(t, "roctracer") || strstr(t, "rocprof")) && !getenv("TETRIS_DIRECT_UNDER_TOOLS"))
                return bad("a profiling tool intercepts the HSA queues");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, hip_device) != hipSuccess) return bad("hipGetDeviceProperties failed");
    if (hsa_init() != HSA_STATUS_SUCCESS) return bad("hsa_init failed");
    FindCtx fc{d, (uint32_t)(((uint32_t)prop.pciBusID << 8) | ((uint32_t)prop.pciDeviceID << 3)), (uint32_t)prop.pciDomainID};
    (void)hsa_iterate_agents(on_agent, &fc);
    if (!fc.found_gpu || !fc.found_cpu) return bad("no HSA agent with the HIP device's PCI address");
    (void)hsa_amd_agent_iterate_memory_pools(d->gpu, on_gpu_pool, d);
    if (d->dev_pool.handle == 0) return bad("no coarse-grained device memory pool");
    (void)hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &d->ts_freq);
    if (!d->ts_freq) return bad("no timestamp frequency");
    std::string why;
    if (!own_code_objects(d->images, why)) return bad(why);
    for (auto& img : d->images) {
        hsa_code_object_reader_t reader;
        hsa_executable_t exe;
        if (hsa_code_object_reader_create_from_memory(img.data(), img.size(), &reader) != HSA_STATUS_SUCCESS) continue;
        if (hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe) != HSA_STATUS_SUCCESS) continue;
        if (hsa_executable_load_agent_code_object(exe, d->gpu, reader, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
            hsa_executable_freeze(exe, nullptr) != HSA_STATUS_SUCCESS) { (void)hsa_executable_destroy(exe); continue; }
        d->exes.push_back(exe);
        SymCtx sc{d};
        (void)hsa_executable_iterate_agent_symbols(exe, d->gpu, on_symbol, &sc);
    }
    if (!d->chain1.ok || !d->duo.ok) return bad("the chained kernels were not found in the loaded code objects");
    if (d->chain1.priv || d->duo.priv) return bad("the chained kernels use scratch memory");
    const uint32_t explicit_bytes = (uint32_t)((sizeof(KArgs) + 7) & ~(size_t)7);
    if (d->chain1.kernarg < explicit_bytes || d->duo.kernarg < explicit_bytes || d->chain1.kernarg > 1024 || d->duo.kernarg > 1024)
        return bad("unexpected kernel-argument segment size");
    d->ok = true;
    return d;
}

static void destroy_queues(Queues& qs) {
    for (int k = 0; k < CHAIN_STREAMS; k++) {
        if (qs.q[k]) (void)hsa_queue_destroy(qs.q[k]);
        if (qs.kernarg[k]) (void)hsa_amd_memory_pool_free(qs.kernarg[k]);
        for (int s = 0; s < 2; s++) if (qs.gate[k][s].handle) (void)hsa_signal_destroy(qs.gate[k][s]);
        if (qs.done[k].handle) (void)hsa_signal_destroy(qs.done[k]);
    }
    if (qs.first.handle) (void)hsa_signal_destroy(qs.first);
    for (int k = 0; k < CHAIN_STREAMS; k++) { qs.q[k] = nullptr; qs.kernarg[k] = nullptr; qs.done[k] = hsa_signal_t{}; qs.gate[k][0] = qs.gate[k][1] = hsa_signal_t{}; }
    qs.first = hsa_signal_t{}; qs.ok = false; qs.affine_ok = false; qs.calibrated = false;
}

static bool make_queues(Device* d, std::string& why) {
    std::lock_guard<std::mutex> lock(g_mutex);
    Queues& qs = d->qs;
    const int depth = CHAIN_STREAMS;
    if (qs.ok) return true;
    for (int k = 0; k < CHAIN_STREAMS; k++) qs.issued[k].store(0);
    qs.slot_bytes = ((std::max(d->chain1.kernarg, d->duo.kernarg) + 63u) & ~63u) + 64u;
    for (int k = 0; k < depth; k++) {
        if (hsa_queue_create(d->gpu, QUEUE_PACKETS, HSA_QUEUE_TYPE_MULTI, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &qs.q[k]) != HSA_STATUS_SUCCESS) { why = "hsa_queue_create failed"; destroy_queues(qs); return false; }
        (void)hsa_amd_profiling_set_profiler_enabled(qs.q[k], 1);
        {   // (experiment knob TETRIS_DIRECT_PRIO=high: no effect that survives swapping the order in which the batches of a process
            // are created — profiles/r03/direct_dispatch.txt)
            const char* e = getenv("TETRIS_DIRECT_PRIO");
            if (e && !strcmp(e, "high")) (void)hsa_amd_queue_set_priority(qs.q[k], HSA_AMD_QUEUE_PRIORITY_HIGH);
        }
        if (hsa_amd_memory_pool_allocate(d->dev_pool, (size_t)SLOTS * qs.slot_bytes + 4096, 0, (void**)&qs.kernarg[k]) != HSA_STATUS_SUCCESS ||
            hsa_amd_agents_allow_access(1, &d->cpu, nullptr, qs.kernarg[k]) != HSA_STATUS_SUCCESS) { why = "no host-visible device memory for the kernel arguments"; destroy_queues(qs); return false; }
        for (int s = 0; s < 2; s++)
            if (hsa_signal_create(0, 0, nullptr, &qs.gate[k][s]) != HSA_STATUS_SUCCESS) { why = "hsa_signal_create failed"; destroy_queues(qs); return false; }
        if (hsa_signal_create(0, 0, nullptr, &qs.done[k]) != HSA_STATUS_SUCCESS) { why = "hsa_signal_create failed"; destroy_queues(qs); return false; }
    }
    if (hsa_signal_create(0, 0, nullptr, &qs.first) != HSA_STATUS_SUCCESS) { why = "hsa_signal_create failed"; destroy_queues(qs); return false; }
    qs.ok = true;
    return true;
}

// hidden arguments of code object v5, behind the explicit ones (8-byte aligned): block counts, group sizes, remainders, ...,
// global offsets at +40, grid dimensions at +64; everything this library's kernels do not use stays zero
static void fill_hidden(char* slot, uint32_t explicit_bytes, uint32_t slot_bytes, uint32_t blocks, uint32_t group) {
    char* h = slot + explicit_bytes;
    const uint32_t room = slot_bytes - explicit_bytes;
    if (room < 72) return;
    const uint32_t bc[3] = {blocks, 1u, 1u};
    const uint16_t gs[3] = {(uint16_t)group, 1, 1};
    const uint16_t dims = 1;
    memcpy(h + 0, bc, 12);
    memcpy(h + 12, gs, 6);
    memcpy(h + 64, &dims, 2);
}

static inline bool wait_signal(hsa_signal_t s) {
    // (active wait: a blocked wait is woken through an interrupt, 10-20 us late)
    for (int round = 0; round < 4000; round++)
        if (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, 2000000, HSA_WAIT_STATE_ACTIVE) < 1) return true;
    return hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) < 1;
}

// Packets are written first and made known to the GPU later, several at a time: `Pending` remembers what has been written
// and not yet rung in.
struct Pending {
    uint64_t written[CHAIN_STREAMS] = {}, rung[CHAIN_STREAMS] = {};
    bool wrote[CHAIN_STREAMS] = {};
    char* last_slot = nullptr;         // kernel arguments written since the last fence
};
static inline void ring(Queues& qs, Pending& pd) {
    // the arguments went through the BAR as write-combined stores: fence, then a read from the device behind them (a PCIe read does
    // not pass the writes in front of it), and only then the doorbells
    if (pd.last_slot) { _mm_sfence(); (void)*(volatile uint32_t*)pd.last_slot; pd.last_slot = nullptr; }
    for (int k = 0; k < CHAIN_STREAMS; k++)
        if (pd.wrote[k] && pd.written[k] != pd.rung[k]) {
            hsa_signal_store_screlease(qs.q[k]->doorbell_signal, (hsa_signal_value_t)(pd.written[k] - 1));
            pd.rung[k] = pd.written[k];
        }
}
static inline char* claim_packet(Queues& qs, Pending& pd, int k) {
    hsa_queue_t* q = qs.q[k];
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
    while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) ring(qs, pd);       // (not reached within the flow-control bounds)
    pd.written[k] = idx + 1; pd.wrote[k] = true;
    return (char*)q->base_address + (idx & (q->size - 1)) * 64;
}
// one kernel launch of `blocks` workgroups of 64 lanes on queue k: arguments into the queue's next slot, then the packet (its header last)
static inline void write_dispatch(Queues& qs, Pending& pd, int k, const Kernel& kern, const void* args, size_t arg_bytes, uint32_t blocks,
                                  int acquire, int release, hsa_signal_t signal) {
    char* slot = qs.kernarg[k] + (size_t)(qs.issued[k].fetch_add(1) % SLOTS) * qs.slot_bytes;
    memcpy(slot, args, arg_bytes);
    fill_hidden(slot, (uint32_t)((arg_bytes + 7) & ~(size_t)7), qs.slot_bytes, blocks, 64);
    pd.last_slot = slot;
    char* p = claim_packet(qs, pd, k);
    hsa_kernel_dispatch_packet_t* d = (hsa_kernel_dispatch_packet_t*)p;
    d->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    d->workgroup_size_x = 64; d->workgroup_size_y = 1; d->workgroup_size_z = 1; d->reserved0 = 0;
    d->grid_size_x = blocks * 64u; d->grid_size_y = 1; d->grid_size_z = 1;
    d->private_segment_size = kern.priv; d->group_segment_size = kern.group;
    d->kernel_object = kern.object; d->kernarg_address = slot; d->reserved2 = 0; d->completion_signal = signal;
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                       (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
}
// a packet that does nothing but complete `signal` once everything in front of it on queue k has retired
static inline void write_barrier(Queues& qs, Pending& pd, int k, hsa_signal_t signal, int scope = HSA_FENCE_SCOPE_SYSTEM) {
    char* p = claim_packet(qs, pd, k);
    memset(p + 2, 0, 62);
    ((hsa_barrier_and_packet_t*)p)->completion_signal = signal;
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                       (scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
}

// Where do a queue's workgroups land?  1024 workgroups (the grid of a 64k-board launch) on every queue, twice: XCD-affine launches are
// enabled only if every launch dealt its blocks round-robin over the eight XCDs — block b on XCD (x0 + b) mod 8 — from the same
// start x0 both times (k_chain_affine checks the same thing again, per workgroup, in every launch).
static void calibrate_affine(Device* d) {
    std::lock_guard<std::mutex> lock(g_mutex);
    Queues& qs = d->qs;
    if (!qs.ok || qs.calibrated) return;
    qs.calibrated = true;
    { const char* e = getenv("TETRIS_AFFINE"); if (e && e[0] == '0') return; }
    if (!d->xcc_probe.ok || !d->chain1_affine.ok || d->xcc_probe.priv || d->chain1_affine.priv || (d->duo_affine.ok && d->duo_affine.priv)) return;
    constexpr uint32_t BLOCKS = 1024;
    const size_t words = (size_t)2 * CHAIN_STREAMS * BLOCKS;
    uint32_t* out = nullptr;
    if (hsa_amd_memory_pool_allocate(d->dev_pool, words * 4, 0, (void**)&out) != HSA_STATUS_SUCCESS) return;
    std::vector<uint32_t> host(words, 0xFFu);
    bool ok = hsa_memory_copy(out, host.data(), words * 4) == HSA_STATUS_SUCCESS;
    for (int rep = 0; rep < 2 && ok; rep++) {
        Pending pd;
        for (int k = 0; k < CHAIN_STREAMS; k++) {
            struct { uint32_t* out; } args = {out + ((size_t)rep * CHAIN_STREAMS + k) * BLOCKS};
            hsa_signal_store_relaxed(qs.done[k], 1);
            write_dispatch(qs, pd, k, d->xcc_probe, &args, sizeof args, BLOCKS, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM, qs.done[k]);
        }
        ring(qs, pd);
        for (int k = 0; k < CHAIN_STREAMS; k++) ok = wait_signal(qs.done[k]) && ok;
    }
    ok = ok && hsa_memory_copy(host.data(), out, words * 4) == HSA_STATUS_SUCCESS;
    (void)hsa_amd_memory_pool_free(out);
    if (!ok) return;
    for (int k = 0; k < CHAIN_STREAMS; k++) {
        const uint32_t* a = &host[(size_t)k * BLOCKS];
        const uint32_t* b = &host[((size_t)CHAIN_STREAMS + k) * BLOCKS];
        if (a[0] > 7u || a[0] != b[0]) return;
        for (uint32_t i = 0; i < BLOCKS; i++)
            if (a[i] != ((a[0] + i) & 7u) || b[i] != a[i]) return;
        qs.xcd_base[k] = a[0];
    }
    qs.affine_ok = true;
}

// waits until queue k has retired everything it was given (before the queues go away; after a test's idle kernel)
static inline void quiesce(Queues& qs) {
    if (!qs.ok) return;
    Pending pd;
    for (int k = 0; k < CHAIN_STREAMS; k++)
        if (qs.q[k]) { hsa_signal_store_relaxed(qs.done[k], 1); write_barrier(qs, pd, k, qs.done[k]); }
    ring(qs, pd);
    for (int k = 0; k < CHAIN_STREAMS; k++)
        if (qs.q[k]) (void)wait_signal(qs.done[k]);
}

}  // namespace aql
