// extern "C" layer, part 4: the per-kernel seam (NTT, dyadic products, RNS stages on raw device buffers; include/sealhip.h)
#include "capi_common.h"

extern "C"
{
    // ------------------------------------------------------------------ per-kernel seam
    SHL_FUNC shl_ntt_forward(void *context, uint64_t *data, uint64_t polys, uint64_t comps, uint64_t first_prime, int lazy, void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(data, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        if (first_prime + comps > c->pool_primes().size())
            throw std::out_of_range("first_prime + comps");
        NttBatch b{};
        b.data = data;
        b.outer_stride = (size_t)comps * c->n();
        b.ncomp = (unsigned)comps;
        b.nouter = (unsigned)polys;
        b.prime_first = (unsigned)first_prime;
        hip_ok(ntt_forward(c->ntt_tables(), b, lazy, (hipStream_t)stream), "ntt_forward");
        SHL_CATCH
    }
    SHL_FUNC shl_ntt_inverse(void *context, uint64_t *data, uint64_t polys, uint64_t comps, uint64_t first_prime, int lazy, void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(data, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        if (first_prime + comps > c->pool_primes().size())
            throw std::out_of_range("first_prime + comps");
        NttBatch b{};
        b.data = data;
        b.outer_stride = (size_t)comps * c->n();
        b.ncomp = (unsigned)comps;
        b.nouter = (unsigned)polys;
        b.prime_first = (unsigned)first_prime;
        hip_ok(ntt_inverse(c->ntt_tables(), b, lazy, (hipStream_t)stream), "ntt_inverse");
        SHL_CATCH
    }
    SHL_FUNC shl_dyadic_product(
        void *context, const uint64_t *a, const uint64_t *b, uint64_t *r, uint64_t polys, uint64_t comps, uint64_t first_prime,
        void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(a, SHL_E_POINTER);
        IfNullRet(b, SHL_E_POINTER);
        IfNullRet(r, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        if (first_prime + comps > c->pool_primes().size())
            throw std::out_of_range("first_prime + comps");
        hip_ok(k_dyadic(c->dev_mods(), a, b, r, (unsigned)c->log_n(), (unsigned)comps, (unsigned)first_prime, polys, (hipStream_t)stream), "dyadic");
        SHL_CATCH
    }
    SHL_FUNC shl_apply_galois(
        void *context, uint64_t chain_index, int ntt_form, uint32_t galois_elt, const uint64_t *in, uint64_t *out, uint64_t polys,
        void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(in, SHL_E_POINTER);
        IfNullRet(out, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        auto l = c->level_by_chain_index(chain_index);
        if (!l)
            throw std::out_of_range("chain_index");
        if (!(galois_elt & 1) || galois_elt >= 2 * c->n())
            throw std::invalid_argument("Galois element is not valid");
        if (in == out)
            throw std::invalid_argument("result cannot point to the same value as operand");
        PlaneGeom g{ (unsigned)c->log_n(), l->K, (unsigned)polys };
        hip_ok(k_apply_galois(c->dev_mods(), in, out, galois_elt, ntt_form, g, 1, (hipStream_t)stream), "apply_galois");
        SHL_CATCH
    }
    SHL_FUNC shl_rns_stage(void *context, uint64_t chain_index, int which, const uint64_t *in, uint64_t *out, uint64_t polys, void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(in, SHL_E_POINTER);
        IfNullRet(out, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        auto l = c->level_by_chain_index(chain_index);
        if (!l)
            throw std::out_of_range("chain_index");
        hipStream_t s = (hipStream_t)stream;
        const unsigned n_log = (unsigned)c->log_n();
        if (which >= 0 && which <= 3)
        {
            if (c->scheme() != Scheme::bfv)
                throw std::logic_error("BEHZ stages exist only for BFV contexts");
            hip_ok(k_behz_stage(c->dev_mods(), l->dev, which, in, out, n_log, polys, s), "behz stage");
        }
        else if (which == 4)
        {
            if (l->K < 2)
                throw std::invalid_argument("level has a single modulus");
            hip_ok(k_bfv_modswitch(c->dev_mods(), l->dev, in, out, n_log, polys, s), "divide_and_round_q_last");
        }
        else if (which == 5)
        {
            if (l->K < 2)
                throw std::invalid_argument("level has a single modulus");
            const unsigned K = l->K;
            const size_t N = c->n();
            Scratch copy(polys * K * N), tt(polys * (K - 1) * N);
            hip_ok(hipMemcpyAsync(copy.p, in, polys * K * N * 8, hipMemcpyDeviceToDevice, s), "copy");
            uint64_t *last = copy.p + (size_t)(K - 1) * N;
            NttBatch bi{};
            bi.data = last;
            bi.outer_stride = (size_t)K * N;
            bi.ncomp = 1;
            bi.nouter = (unsigned)polys;
            bi.prime_first = K - 1;
            hip_ok(ntt_inverse(c->ntt_tables(), bi, 0, s), "intt last");
            NttBatch b{};
            b.data = tt.p;
            b.outer_stride = (size_t)(K - 1) * N;
            b.ncomp = K - 1;
            b.nouter = (unsigned)polys;
            b.src = last;
            b.src_outer_stride = (size_t)K * N;
            b.src_ncomp = 1;
            b.src_mode = 2;
            b.src_half = l->dev.half_q_last;
            b.src_q = l->dev.q_last;
            b.src_fix = l->dev.round_fix;
            hip_ok(ntt_forward(c->ntt_tables(), b, 1, s), "ntt correction");
            hip_ok(k_rescale_combine(c->dev_mods(), l->dev.inv_q_last_mod_q, copy.p, tt.p, out, n_log, K, polys, s), "combine");
            hip_ok(hipStreamSynchronize(s), "sync");
        }
        else
            throw std::invalid_argument("unknown stage");
        SHL_CATCH
    }
    SHL_FUNC SealHip_SetStagedHostCopies(bool enabled)
    {
        set_staged_host_copies(enabled);
        return SHL_S_OK;
    }
    SHL_FUNC SealHip_ReleasePool(void)
    {
        SHL_TRY
        DevicePool::global().release_all();
        SHL_CATCH
    }
    SHL_FUNC SealHip_PoolStats(uint64_t *bytes_held, uint64_t *cross_stream_waits)
    {
        SHL_TRY
        if (bytes_held)
            *bytes_held = DevicePool::global().bytes_held();
        if (cross_stream_waits)
            *cross_stream_waits = DevicePool::global().cross_stream_waits();
        SHL_CATCH
    }
    SHL_FUNC SealHip_KsChunkStats(uint64_t *calls, uint64_t *chunks, uint64_t *scratch_bytes_max)
    {
        SHL_TRY
        uint64_t w = 0;
        ks_chunk_stats(calls, chunks, &w);
        if (scratch_bytes_max)
            *scratch_bytes_max = w * 8;
        SHL_CATCH
    }
    SHL_FUNC SealHip_ProductStats(uint64_t *fused, uint64_t *formed, uint64_t *dropped)
    {
        SHL_TRY
        uint64_t f, p, d;
        lazy_product_stats(f, p, d);
        if (fused)
            *fused = f;
        if (formed)
            *formed = p;
        if (dropped)
            *dropped = d;
        SHL_CATCH
    }
    SHL_FUNC SealHip_GaloisStats(uint64_t *gathered, uint64_t *permuted)
    {
        SHL_TRY
        uint64_t g, p;
        galois_path_stats(g, p);
        if (gathered)
            *gathered = g;
        if (permuted)
            *permuted = p;
        SHL_CATCH
    }
    SHL_FUNC SealHip_TailStats(uint64_t *folded, uint64_t *plain, uint64_t *dropped)
    {
        SHL_TRY
        uint64_t f, p, d;
        lazy_tail_stats(f, p, d);
        if (folded)
            *folded = f;
        if (plain)
            *plain = p;
        if (dropped)
            *dropped = d;
        SHL_CATCH
    }
    SHL_FUNC shl_device_count(int *count)
    {
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        *count = 0;
        if (hipGetDeviceCount(count) != hipSuccess)
            *count = 0;
        SHL_CATCH
    }
    SHL_FUNC shl_set_device(int device)
    {
        SHL_TRY
        hip_ok(hipSetDevice(device), "hipSetDevice");
        SHL_CATCH
    }
    SHL_FUNC shl_stream_create(bool non_blocking, void **hip_stream)
    {
        IfNullRet(hip_stream, SHL_E_POINTER);
        SHL_TRY
        hipStream_t s = nullptr;
        hip_ok(hipStreamCreateWithFlags(&s, non_blocking ? hipStreamNonBlocking : hipStreamDefault), "hipStreamCreateWithFlags");
        *hip_stream = s;
        SHL_CATCH
    }
    SHL_FUNC shl_stream_destroy(void *hip_stream)
    {
        SHL_TRY
        hip_ok(hipStreamSynchronize((hipStream_t)hip_stream), "hipStreamSynchronize");
        hip_ok(hipStreamDestroy((hipStream_t)hip_stream), "hipStreamDestroy");
        SHL_CATCH
    }
    SHL_FUNC shl_malloc(uint64_t bytes, void **device_ptr)
    {
        IfNullRet(device_ptr, SHL_E_POINTER);
        SHL_TRY
        if (hipMalloc(device_ptr, bytes) != hipSuccess)
            throw std::bad_alloc();
        SHL_CATCH
    }
    SHL_FUNC shl_free(void *device_ptr)
    {
        SHL_TRY
        hip_ok(hipFree(device_ptr), "hipFree");
        SHL_CATCH
    }
    SHL_FUNC shl_memcpy_h2d(void *device_dst, const void *host_src, uint64_t bytes)
    {
        SHL_TRY
        hip_ok(hipMemcpy(device_dst, host_src, bytes, hipMemcpyHostToDevice), "H2D");
        SHL_CATCH
    }
    SHL_FUNC shl_memcpy_d2h(void *host_dst, const void *device_src, uint64_t bytes)
    {
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "sync");
        hip_ok(hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost), "D2H");
        SHL_CATCH
    }
    SHL_FUNC shl_device_synchronize(void)
    {
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
        SHL_CATCH
    }
    SHL_FUNC shl_timer_create(void **timer)
    {
        IfNullRet(timer, SHL_E_POINTER);
        SHL_TRY
        auto t = new Timer();
        hip_ok(hipEventCreate(&t->e0), "hipEventCreate");
        hip_ok(hipEventCreate(&t->e1), "hipEventCreate");
        *timer = t;
        SHL_CATCH
    }
    SHL_FUNC shl_timer_destroy(void *timer)
    {
        IfNullRet(timer, SHL_E_POINTER);
        auto t = as<Timer>(timer);
        (void)hipEventDestroy(t->e0);
        (void)hipEventDestroy(t->e1);
        delete t;
        return SHL_S_OK;
    }
    SHL_FUNC shl_timer_start(void *timer, void *stream)
    {
        IfNullRet(timer, SHL_E_POINTER);
        SHL_TRY
        hip_ok(hipEventRecord(as<Timer>(timer)->e0, (hipStream_t)stream), "hipEventRecord");
        SHL_CATCH
    }
    SHL_FUNC shl_timer_stop(void *timer, void *stream, float *milliseconds)
    {
        IfNullRet(timer, SHL_E_POINTER);
        IfNullRet(milliseconds, SHL_E_POINTER);
        SHL_TRY
        auto t = as<Timer>(timer);
        hip_ok(hipEventRecord(t->e1, (hipStream_t)stream), "hipEventRecord");
        hip_ok(hipEventSynchronize(t->e1), "hipEventSynchronize");
        hip_ok(hipEventElapsedTime(milliseconds, t->e0, t->e1), "hipEventElapsedTime");
        SHL_CATCH
    }
}
