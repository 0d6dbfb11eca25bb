// eb_prims.hip -- batched rank / access / LF / FL kernels over the packed BWT.
// Replaces dna_string::parallel_rank (dna_string.hpp:140-152), operator[] (:113-135),
// dna_bwt::LF(range_t) (dna_bwt.hpp:138-166), LF(range_t,char) (:168-192), FL (:115-133),
// LF(sa_node) (:323-356).
#include "eb_internal.h"

// ------------------------------------------------------------------ kernels

// K1: rank4_batch.  Four adjacent lanes serve one query: lane j loads the j-th 16-byte chunk of the
// query's 64-byte block (one coalesced segment per query, 16 queries per wave-instruction), counts
// all four letters in its 32 symbols, and after two lane swaps writes the rank of letter j.
// Grid-stride so that the launch stays at a few thousand workgroups for any m.
__global__ void __launch_bounds__(256) k_rank4_batch(eb_dev_index x, const uint64_t* __restrict__ pos, uint64_t m,
                                                     uint64_t* __restrict__ out4) {
    const uint32_t j = threadIdx.x & 3u;
    uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 2;
    // Two queries in flight per lane group to cover HBM latency.
    for (; q < m; q += 2 * stride) {
        uint64_t q1 = q + stride;
        bool has1 = q1 < m; // uniform inside a quad
        // positions past the end count as the end: the host form has checked them (check_positions), the device form
        // (eb_rank4_batch_device) takes them as they are, and a block index out of range would be a GPU fault
        uint64_t i0 = pos[q];
        uint64_t i1 = has1 ? pos[q1] : i0;
        i0 = i0 < x.n ? i0 : x.n;
        i1 = i1 < x.n ? i1 : x.n;
        uint4 c0 = eb_load_chunk(x.chunks + ((i0 >> 7) << 2) + j);
        uint4 c1 = eb_load_chunk(x.chunks + ((i1 >> 7) << 2) + j);
        uint32_t p0 = eb_chunk_counts(c0, eb_prefix_mask32((uint32_t)(i0 & 127u), j));
        uint32_t p1 = eb_chunk_counts(c1, eb_prefix_mask32((uint32_t)(i1 & 127u), j));
        p0 = eb_quad_sum(p0);
        p1 = eb_quad_sum(p1);
        out4[q * 4 + j] = x.super[((i0 >> EB_SUPER_SHIFT) << 2) + j] + c0.w + ((p0 >> (8u * j)) & 0xFFu);
        if (has1) out4[q1 * 4 + j] = x.super[((i1 >> EB_SUPER_SHIFT) << 2) + j] + c1.w + ((p1 >> (8u * j)) & 0xFFu);
    }
}

__global__ void __launch_bounds__(256) k_access_batch(eb_dev_index x, const uint64_t* __restrict__ pos, uint64_t m,
                                                      uint8_t* __restrict__ out) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    out[q] = (uint8_t)eb_code_to_ascii(eb_code(x, pos[q]), x.term);
}

// K2: LF(range): quad per range, second rank skipped when the range is empty (dna_bwt.hpp:148-151).
__global__ void __launch_bounds__(256) k_lf_range_batch(eb_dev_index x, const uint64_t* __restrict__ lr, uint64_t m,
                                                        uint64_t* __restrict__ out8) {
    const uint32_t j = threadIdx.x & 3u;
    uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    if (q >= m) return; // uniform inside a quad
    uint64_t l = lr[2 * q], r = lr[2 * q + 1];
    uint64_t a = eb_rank4_quad(x, l, j);
    uint64_t b = eb_rank4_quad(x, r > l ? r : l, j);
    out8[8 * q + 2 * j] = x.F[j] + a;
    out8[8 * q + 2 * j + 1] = x.F[j] + b;
}

__global__ void __launch_bounds__(256) k_lf_range_sym_batch(eb_dev_index x, const uint64_t* __restrict__ lr,
                                                            const uint8_t* __restrict__ sym, uint64_t m,
                                                            uint64_t* __restrict__ out2) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    uint64_t l = lr[2 * q], r = lr[2 * q + 1];
    uint8_t s = sym[q];
    int c = s == 'A' ? 0 : s == 'C' ? 1 : s == 'G' ? 2 : s == 'T' ? 3 : -1;
    if (c < 0) { out2[2 * q] = 0; out2[2 * q + 1] = 0; return; } // LF(range,c) returns {} for other symbols (dna_bwt.hpp:190)
    rank4_t a = eb_rank4(x, l);
    rank4_t b = r > l ? eb_rank4(x, r) : a;
    out2[2 * q] = x.F[c] + a.v[c];
    out2[2 * q + 1] = x.F[c] + b.v[c];
}

__global__ void __launch_bounds__(256) k_fl_batch(eb_dev_index x, const uint64_t* __restrict__ pos, uint64_t m,
                                                  uint64_t* __restrict__ out) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    uint64_t i = pos[q];
    int c = eb_F_letter(x, i);
    out[q] = c < 0 ? ~0ull : eb_FL(x, i, c);
}

// LF(sa_node): one rank per distinct boundary (dna_bwt.hpp:332-347)
__global__ void __launch_bounds__(256) k_lf_node_batch(eb_dev_index x, const uint64_t* __restrict__ nodes7, uint64_t m,
                                                       uint64_t* __restrict__ out28) {
    uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const uint64_t* N = nodes7 + 7 * q;
    rank4_t r[6];
    r[0] = eb_rank4(x, N[0]);
    for (int k = 1; k < 6; ++k) r[k] = N[k] == N[k - 1] ? r[k - 1] : eb_rank4(x, N[k]);
    for (int c = 0; c < 4; ++c) {
        for (int k = 0; k < 6; ++k) out28[28 * q + 7 * c + k] = x.F[c] + r[k].v[c];
        out28[28 * q + 7 * c + 6] = N[6] + 1;
    }
}

// ------------------------------------------------------------------ host wrappers

namespace {

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) { eb_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return EB_ERR_NOMEM; }
        return EB_OK;
    }
};

int check_handle(const eb_index* h) {
    if (!h) { eb_set_error("null index handle"); return EB_ERR_ARG; }
    EB_HIP(hipSetDevice(h->device));
    return EB_OK;
}

unsigned grid_for(uint64_t threads) { return (unsigned)((threads + 255) / 256); }

// rank grid: 4 lanes per query, 2 queries per lane group per trip, capped so the kernel grid-strides
unsigned rank_grid(uint64_t m) {
    uint64_t blocks = (m * 4 + 255) / 256;
    const uint64_t cap = 256ull * 16; // 16 workgroups per CU
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

template <typename Launch>
int run_batch(eb_index* h, const void* in, size_t in_bytes, const void* in2, size_t in2_bytes, void* out, size_t out_bytes,
              Launch launch) {
    int rc = check_handle(h);
    if (rc != EB_OK) return rc;
    DevBuf din, din2, dout;
    if ((rc = din.alloc(in_bytes)) != EB_OK || (rc = dout.alloc(out_bytes)) != EB_OK) return rc;
    if (in2 && (rc = din2.alloc(in2_bytes)) != EB_OK) return rc;
    EB_HIP(hipMemcpyAsync(din.p, in, in_bytes, hipMemcpyHostToDevice, h->stream));
    if (in2) EB_HIP(hipMemcpyAsync(din2.p, in2, in2_bytes, hipMemcpyHostToDevice, h->stream));
    launch(din.p, din2.p, dout.p);
    EB_HIP(hipGetLastError());
    EB_HIP(hipMemcpyAsync(out, dout.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
    EB_HIP(hipStreamSynchronize(h->stream));
    return EB_OK;
}

int check_positions(const eb_index* h, const uint64_t* pos, uint64_t m, bool allow_n, const char* what) {
    for (uint64_t q = 0; q < m; ++q)
        if (pos[q] > h->n || (!allow_n && pos[q] == h->n)) {
            eb_set_error("%s: position %llu out of range (n = %llu)", what, (unsigned long long)pos[q], (unsigned long long)h->n);
            return EB_ERR_ARG;
        }
    return EB_OK;
}

}  // namespace

extern "C" int eb_rank4_batch(eb_index* h, const uint64_t* pos, uint64_t m, uint64_t* out4) {
    if (m == 0) return check_handle(h);
    if (!h || !pos || !out4) { eb_set_error("null argument"); return EB_ERR_ARG; }
    int rc = check_positions(h, pos, m, true, "eb_rank4_batch");
    if (rc != EB_OK) return rc;
    eb_dev_index x = h->dev();
    return run_batch(h, pos, m * 8, nullptr, 0, out4, m * 32, [&](void* a, void*, void* o) {
        hipLaunchKernelGGL(k_rank4_batch, dim3(rank_grid(m)), dim3(256), 0, h->stream, x, (const uint64_t*)a, m, (uint64_t*)o);
    });
}

extern "C" int eb_rank4_batch_device(eb_index* h, const void* d_pos, uint64_t m, void* d_out4, float* kernel_ms) {
    int rc = check_handle(h);
    if (rc != EB_OK) return rc;
    if (m == 0) { if (kernel_ms) *kernel_ms = 0.f; return EB_OK; }
    if (!d_pos || !d_out4) { eb_set_error("null argument"); return EB_ERR_ARG; }
    eb_dev_index x = h->dev();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (kernel_ms) {
        EB_HIP(hipEventCreate(&e0));
        EB_HIP(hipEventCreate(&e1));
        EB_HIP(hipEventRecord(e0, h->stream));
    }
    hipLaunchKernelGGL(k_rank4_batch, dim3(rank_grid(m)), dim3(256), 0, h->stream, x, (const uint64_t*)d_pos, m, (uint64_t*)d_out4);
    EB_HIP(hipGetLastError());
    if (kernel_ms) {
        EB_HIP(hipEventRecord(e1, h->stream));
        EB_HIP(hipEventSynchronize(e1));
        EB_HIP(hipEventElapsedTime(kernel_ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    } else {
        EB_HIP(hipStreamSynchronize(h->stream));
    }
    return EB_OK;
}

extern "C" int eb_access_batch(eb_index* h, const uint64_t* pos, uint64_t m, uint8_t* out) {
    if (m == 0) return check_handle(h);
    if (!h || !pos || !out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    int rc = check_positions(h, pos, m, false, "eb_access_batch");
    if (rc != EB_OK) return rc;
    eb_dev_index x = h->dev();
    return run_batch(h, pos, m * 8, nullptr, 0, out, m, [&](void* a, void*, void* o) {
        hipLaunchKernelGGL(k_access_batch, dim3(grid_for(m)), dim3(256), 0, h->stream, x, (const uint64_t*)a, m, (uint8_t*)o);
    });
}

extern "C" int eb_lf_range_batch(eb_index* h, const uint64_t* lr, uint64_t m, uint64_t* out8) {
    if (m == 0) return check_handle(h);
    if (!h || !lr || !out8) { eb_set_error("null argument"); return EB_ERR_ARG; }
    int rc = check_positions(h, lr, 2 * m, true, "eb_lf_range_batch");
    if (rc != EB_OK) return rc;
    eb_dev_index x = h->dev();
    return run_batch(h, lr, m * 16, nullptr, 0, out8, m * 64, [&](void* a, void*, void* o) {
        hipLaunchKernelGGL(k_lf_range_batch, dim3(grid_for(m * 4)), dim3(256), 0, h->stream, x, (const uint64_t*)a, m, (uint64_t*)o);
    });
}

extern "C" int eb_lf_range_sym_batch(eb_index* h, const uint64_t* lr, const uint8_t* sym, uint64_t m, uint64_t* out2) {
    if (m == 0) return check_handle(h);
    if (!h || !lr || !sym || !out2) { eb_set_error("null argument"); return EB_ERR_ARG; }
    int rc = check_positions(h, lr, 2 * m, true, "eb_lf_range_sym_batch");
    if (rc != EB_OK) return rc;
    eb_dev_index x = h->dev();
    return run_batch(h, lr, m * 16, sym, m, out2, m * 16, [&](void* a, void* b, void* o) {
        hipLaunchKernelGGL(k_lf_range_sym_batch, dim3(grid_for(m)), dim3(256), 0, h->stream, x, (const uint64_t*)a, (const uint8_t*)b, m, (uint64_t*)o);
    });
}

extern "C" int eb_fl_batch(eb_index* h, const uint64_t* pos, uint64_t m, uint64_t* out) {
    if (m == 0) return check_handle(h);
    if (!h || !pos || !out) { eb_set_error("null argument"); return EB_ERR_ARG; }
    int rc = check_positions(h, pos, m, false, "eb_fl_batch");
    if (rc != EB_OK) return rc;
    for (uint64_t q = 0; q < m; ++q)
        if (pos[q] < h->F[0]) { eb_set_error("eb_fl_batch: FL is undefined on terminator rows (position %llu)", (unsigned long long)pos[q]); return EB_ERR_ARG; }
    eb_dev_index x = h->dev();
    return run_batch(h, pos, m * 8, nullptr, 0, out, m * 8, [&](void* a, void*, void* o) {
        hipLaunchKernelGGL(k_fl_batch, dim3(grid_for(m)), dim3(256), 0, h->stream, x, (const uint64_t*)a, m, (uint64_t*)o);
    });
}

extern "C" int eb_lf_node_batch(eb_index* h, const uint64_t* nodes7, uint64_t m, uint64_t* out28) {
    if (m == 0) return check_handle(h);
    if (!h || !nodes7 || !out28) { eb_set_error("null argument"); return EB_ERR_ARG; }
    for (uint64_t q = 0; q < m; ++q)
        for (int k = 0; k < 6; ++k)
            if (nodes7[7 * q + k] > h->n || (k && nodes7[7 * q + k] < nodes7[7 * q + k - 1])) {
                eb_set_error("eb_lf_node_batch: node %llu has invalid boundaries", (unsigned long long)q);
                return EB_ERR_ARG;
            }
    eb_dev_index x = h->dev();
    return run_batch(h, nodes7, m * 56, nullptr, 0, out28, m * 224, [&](void* a, void*, void* o) {
        hipLaunchKernelGGL(k_lf_node_batch, dim3(grid_for(m)), dim3(256), 0, h->stream, x, (const uint64_t*)a, m, (uint64_t*)o);
    });
}
