// eb_wire.hip -- the compact wire image of a call set (no reference counterpart: the reference is one process).  What a rank of
// a multi-GPU run hands to rank 0 -- over its own PCIe link into the host's arena, or GPU -> GPU with RCCL -- is not the 80-byte
// per two-allele-cluster image the writer reads but 44 bytes: include/ebwt2indel.h, eb_scan_job_device_wire.
//   k_wire4  tight records (EBCALLS4), sizes read from the JobLayout in device memory: packed by the job itself (EB_SCAN_WIRE)
//   k_wire3  full-width records (EBCALLS3), for the call set whose fields do not fit the tight ones
#include "eb_scan_internal.h"

namespace {

__device__ __forceinline__ void write_header(uint8_t* dst, uint64_t magic, uint64_t count, uint64_t n_chains, int32_t k_left, int32_t k_right) {
    uint64_t* h = reinterpret_cast<uint64_t*>(dst);
    h[0] = magic; h[1] = count; h[2] = n_chains; h[3] = (uint64_t)(uint32_t)k_left; h[4] = (uint64_t)(uint32_t)k_right; h[5] = h[6] = h[7] = 0;
}

// grid-stride over max(clusters, chains, context words); *bad counts fields that do not fit
__global__ void __launch_bounds__(256) k_wire4(const uint8_t* __restrict__ blob, ScanCounters* __restrict__ cnt, uint8_t* __restrict__ wire, uint32_t lw,
                                              uint32_t rw, int32_t k_left, int32_t k_right) {
    const JobLayout L = cnt->lay;
    const uint64_t count = L.ns, n_chains = L.n_chains;
    const eb_cluster* __restrict__ recs = reinterpret_cast<const eb_cluster*>(blob + sizeof(eb_calls_header));
    const eb_chain* __restrict__ chains = reinterpret_cast<const eb_chain*>(blob + L.off_crec);
    eb_wire4_cluster* __restrict__ wc = reinterpret_cast<eb_wire4_cluster*>(wire + sizeof(eb_calls_header));
    eb_wire4_chain* __restrict__ wk = reinterpret_cast<eb_wire4_chain*>(wire + L.wire_off_chain);
    const uint64_t words = n_chains * lw + count * rw; // left words, then right words: contiguous in both images
    const uint64_t* __restrict__ src = reinterpret_cast<const uint64_t*>(blob + L.off_left);
    uint64_t* __restrict__ dst = reinterpret_cast<uint64_t*>(wire + L.wire_off_ctx);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    if (tid == 0) {
        write_header(wire, EB_WIRE4_MAGIC, count, n_chains, k_left, k_right);
        // the bytes between the last chain record and the context words (0 or 4: records are 12 and 4 bytes, the words 8-aligned)
        for (uint64_t o = L.wire_off_chain + n_chains * sizeof(eb_wire4_chain); o < L.wire_off_ctx; ++o) wire[o] = 0;
    }
    uint32_t bad = 0;
    for (uint64_t i = tid; i < count; i += stride) {
        const eb_cluster r = recs[i];
        eb_wire4_cluster w;
        w.begin_lo = (uint32_t)r.begin;
        w.begin_hi = (uint8_t)(r.begin >> 32);
        w.slots_found = (uint8_t)((r.n_slots & 15u) | ((r.right_found ? 1u : 0u) << 4) | (((r.reserved & EB_RIGHT_SKIPPED) ? 1u : 0u) << 5));
        w.len = (uint16_t)(r.end - r.begin);
        w.right_len = (uint16_t)r.right_len;
        w.pad = 0;
        if ((r.begin >> 40) || r.end - r.begin > 0xFFFFull || r.right_len > 0xFFFFu || r.n_slots > 15u) ++bad;
        wc[i] = w;
    }
    for (uint64_t i = tid; i < n_chains; i += stride) {
        const eb_chain k = chains[i];
        const uint32_t letter = k.sym == 'A' ? 0u : k.sym == 'C' ? 1u : k.sym == 'G' ? 2u : 3u;
        eb_wire4_chain w;
        w.side_sym_len = (uint16_t)((k.side & 1u) | (letter << 1) | (k.len << 3));
        w.support = (uint16_t)k.support;
        if (k.len > 8191u || k.support > 0xFFFFu) ++bad;
        wk[i] = w;
    }
    for (uint64_t i = tid; i < words; i += stride) dst[i] = src[i];
    if (bad) atomicAdd(&cnt->wire_bad, (unsigned long long)bad);
}

__global__ void __launch_bounds__(256) k_wire3(const uint8_t* __restrict__ blob, uint64_t count, uint64_t n_chains, uint8_t* __restrict__ wire, uint32_t lw,
                                              uint32_t rw, int32_t k_left, int32_t k_right, unsigned long long* __restrict__ bad) {
    JobLayout L;
    eb_job_layout(L, (uint32_t)count, (uint32_t)n_chains, lw, rw);
    const eb_cluster* __restrict__ recs = reinterpret_cast<const eb_cluster*>(blob + sizeof(eb_calls_header));
    const eb_chain* __restrict__ chains = reinterpret_cast<const eb_chain*>(blob + L.off_crec);
    eb_wire_cluster* __restrict__ wc = reinterpret_cast<eb_wire_cluster*>(wire + sizeof(eb_calls_header));
    eb_wire_chain* __restrict__ wk = reinterpret_cast<eb_wire_chain*>(wire + sizeof(eb_calls_header) + count * sizeof(eb_wire_cluster));
    const uint64_t words = n_chains * lw + count * rw;
    const uint64_t* __restrict__ src = reinterpret_cast<const uint64_t*>(blob + L.off_left);
    uint64_t* __restrict__ dst = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(wk) + n_chains * sizeof(eb_wire_chain));
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    if (tid == 0) write_header(wire, EB_WIRE_MAGIC, count, n_chains, k_left, k_right);
    for (uint64_t i = tid; i < count; i += stride) {
        const eb_cluster r = recs[i];
        eb_wire_cluster w;
        w.begin = r.begin;
        w.len = (uint32_t)(r.end - r.begin);
        w.n_slots = (uint8_t)r.n_slots;
        w.right_found = (uint8_t)((r.right_found ? 1u : 0u) | ((r.reserved & EB_RIGHT_SKIPPED) ? 2u : 0u));
        w.right_len = (uint16_t)r.right_len;
        if (r.end - r.begin > 0xFFFFFFFFull || r.right_len > 0xFFFFu || r.n_slots > 0xFFu) atomicAdd(bad, 1ull);
        wc[i] = w;
    }
    for (uint64_t i = tid; i < n_chains; i += stride) {
        const eb_chain k = chains[i];
        eb_wire_chain w;
        w.side = k.side;
        w.sym = k.sym;
        w.len = (uint16_t)k.len;
        w.support = k.support;
        if (k.len > 0xFFFFu) atomicAdd(bad, 1ull);
        wk[i] = w;
    }
    for (uint64_t i = tid; i < words; i += stride) dst[i] = src[i];
}

unsigned wire_grid(uint64_t most) {
    const uint64_t g = (most + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); // grid-stride beyond: a few thousand workgroups fill the chip
}

}  // namespace

void eb_launch_wire4(hipStream_t st, const uint8_t* blob, ScanCounters* cnt, uint8_t* wire, uint32_t lw, uint32_t rw, int32_t k_left, int32_t k_right,
                     uint64_t cap_ns, uint64_t cap_chains) {
    const uint64_t words = cap_chains * lw + cap_ns * rw;
    hipLaunchKernelGGL(k_wire4, dim3(wire_grid(words)), dim3(256), 0, st, blob, cnt, wire, lw, rw, k_left, k_right);
}

void eb_launch_wire3(hipStream_t st, const uint8_t* blob, uint64_t count, uint64_t n_chains, uint8_t* wire, uint32_t lw, uint32_t rw, int32_t k_left,
                     int32_t k_right, unsigned long long* d_bad) {
    const uint64_t words = n_chains * lw + count * rw;
    hipLaunchKernelGGL(k_wire3, dim3(wire_grid(words > count ? words : count)), dim3(256), 0, st, blob, count, n_chains, wire, lw, rw, k_left, k_right, d_bad);
}

// loads this file's code object now (the runtime does it with the first launch otherwise: milliseconds in front of the first job)
void eb_wire_preload() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_wire3));
}
