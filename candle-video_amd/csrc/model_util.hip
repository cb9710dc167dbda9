#include "model_util.h"
#include "options.h"

int ltx_stage_src(const ltx_weight* w, const void** dev_src, void** temp_to_free) {
    *temp_to_free = nullptr;
    if (w->on_device) { *dev_src = w->data; return LTX_OK; }
    size_t bytes = (size_t)ltx_numel(w) * ltx_dt_size(w->dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32);
    void* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, bytes ? bytes : 16));
    hipError_t e = hipMemcpy(tmp, w->data, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(tmp); ltx_set_error(std::string("hipMemcpy H2D: ") + hipGetErrorString(e)); return LTX_ERR_HIP; }
    *dev_src = tmp; *temp_to_free = tmp;
    return LTX_OK;
}

int ltx_upload_cast(const ltx_weight* w, void* dst, int dst_dtype, int64_t expect_numel, const std::string& name) {
    if (ltx_numel(w) != expect_numel)
        LTX_FAIL(LTX_ERR_ARG, "weight '" + name + "': expected " + std::to_string(expect_numel) + " elements, got " + std::to_string(ltx_numel(w)));
    const void* src = nullptr; void* tmp = nullptr;
    LTX_TRY(ltx_stage_src(w, &src, &tmp));
    int rc = ltx_launch_cast(src, w->dtype == LTX_BF16 ? LTX_DT_BF16 : LTX_DT_F32, dst, dst_dtype, expect_numel, 0);
    hipError_t e = hipDeviceSynchronize();
    if (tmp) (void)hipFree(tmp);
    if (rc != LTX_OK) return rc;
    if (e != hipSuccess) { ltx_set_error(std::string("weight upload: ") + hipGetErrorString(e)); return LTX_ERR_HIP; }
    return LTX_OK;
}

int ltx_load_tensor(const WeightMap& wm, const std::string& name, int64_t numel, int dtype, void** out, bool optional) {
    *out = nullptr;
    const ltx_weight* w = wm.find(name);
    if (!w) {
        if (optional) return LTX_OK;
        LTX_FAIL(LTX_ERR_MISSING_WEIGHT, "missing weight '" + name + "'");
    }
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, (size_t)numel * ltx_dt_size(dtype) + 16));
    int rc = ltx_upload_cast(w, p, dtype, numel, name);
    if (rc != LTX_OK) { (void)hipFree(p); return rc; }
    *out = p;
    return LTX_OK;
}

int ltx_load_linear(const WeightMap& wm, const std::string& prefix, int in, int out, int dtype, LinearW* l) {
    l->in = in; l->out = out;
    LTX_TRY(ltx_load_tensor(wm, prefix + ".weight", (int64_t)in * out, dtype, &l->w));
    LTX_TRY(ltx_load_tensor(wm, prefix + ".bias", out, dtype, &l->b, /*optional=*/true));
    return LTX_OK;
}

int ltx_linear(const LinearW& l, const void* x, int lda, void* y, int ldc, int M, int dtype, int epi, hipStream_t s,
               const void* resid, int ldr, const float* gate, int gate_stride, int rows_per_batch, float* rowsq) {
    GemmArgs g;
    g.A = x; g.W = l.w; g.C = y; g.bias = l.b; g.resid = resid; g.gate = gate;
    g.M = M; g.N = l.out; g.K = l.in; g.lda = lda; g.ldc = ldc; g.ldr = ldr;
    g.rows_per_batch = rows_per_batch; g.gate_stride = gate_stride; g.rowsq = rowsq;
    // Opt-in experiment (x_ring_pack=1 in an experiment build; measured: C1 654 vs 654 frames/s, T5-XXL 4.87 vs 5.08 ms - not worth a second copy
    // of the weights, docs/lab_notes.md R4.11): small-M calls hand gemm_ring.hip a tile-contiguous copy of the weights, built
    // once per layer on its first such call (+ 100 % of the layer's weight bytes).
    if (dtype == LTX_DT_BF16 && M <= 512 && l.out >= 32 && l.out % 4 == 0 && l.in % 8 == 0) {
        const bool pack_on = ltx_exp("ring_pack", 0) == 1;      // experiment builds only
        if (pack_on && !l.wp && !l.wp_tried) {
            l.wp_tried = true;
            void* p = nullptr;
            if (hipMalloc(&p, ltx_ring_packed_bytes(l.out, l.in)) == hipSuccess) {
                std::shared_ptr<void> hold(p, [](void* q) { (void)hipFree(q); });
                if (ltx_pack_ring_weights(l.w, l.out, l.in, p, s) == LTX_OK && hipStreamSynchronize(s) == hipSuccess) l.wp = hold;
            } else (void)hipGetLastError();                 // out of memory: the row-major weights serve
        }
        g.Wp = l.wp.get();
    }
    return ltx_launch_gemm(g, dtype, epi, s);
}

static thread_local bool t_oom = false;
void ltx_note_oom() { t_oom = true; }
bool ltx_take_oom() { const bool v = t_oom; t_oom = false; return v; }
