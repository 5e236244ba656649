/*
 * Drop-in for /root/reference/src/model.c on MI355X: same seven functions (include/model.h), but
 * the session is a set of gfx950 engines (one per GPU) and run_inference() is one hand-written HIP
 * forward instead of g_ort->Run().  Host side is pure C; all device work goes through the C-ABI of
 * include/gliclass_hip.h.  There is no CPU execution path: create_ort_session() fails if no GPU
 * is visible.
 *
 * Runtime settings (the reference has only compile-time #defines, /root/reference/include/configs.h):
 *   GLICLASS_DEVICES = "0,1,.." GPUs a session spans (default "0"; "all" = every visible GPU)
 *   GLICLASS_DTYPE   = f32 | f16 | bf16   arithmetic mode (default f32: fp32 data, every matrix product as split-f16 MFMAs — the
 *                      mode that meets the reference's own 1e-3 tolerance, /root/reference/ONNX_CONVERTING/test_onnx.py:30, on
 *                      every model tried; f16 / bf16 are opt-in throughput modes whose operand rounding can exceed it)
 */
#include "model.h"

#include <omp.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "glc_host_internal.h"
#include "glc_weights.h"

__attribute__((weak)) const OrtApi* g_ort = NULL;

/* /root/reference/src/model.c:17-29 — ragged int** -> contiguous int64 (caller frees) */
int64_t* flatten_int_array(int** data, size_t rows, size_t cols) {
    int64_t* flat = (int64_t*)malloc((rows && cols ? rows * cols : 1) * sizeof(int64_t));
    if (!flat) { fprintf(stderr, "Error: Memory allocation for flat_data failed\n"); return NULL; }
    for (size_t r = 0; r < rows; ++r) {
        const int* src = data[r];
        int64_t* dst = flat + r * cols;
        for (size_t c = 0; c < cols; ++c) dst[c] = (int64_t)src[c];
    }
    return flat;
}

/* /root/reference/src/model.c:39-71 — wrap as an INT64 [rows, cols] tensor (no copy, no ownership) */
OrtValue* create_tensor(int64_t* data, size_t rows, size_t cols) {
    if (!g_ort) { fprintf(stderr, "Error: initialize_ort_api() was not called\n"); return NULL; }
    OrtMemoryInfo* mi = NULL;
    OrtStatus* st = g_ort->CreateCpuMemoryInfo(OrtArenaAllocator, OrtMemTypeDefault, &mi);
    if (st) { fprintf(stderr, "Error: Failed to create MemoryInfo: %s\n", g_ort->GetErrorMessage(st)); g_ort->ReleaseStatus(st); return NULL; }
    int64_t dims[2] = {(int64_t)rows, (int64_t)cols};
    OrtValue* t = NULL;
    st = g_ort->CreateTensorWithDataAsOrtValue(mi, data, rows * cols * sizeof(int64_t), dims, 2, ONNX_TENSOR_ELEMENT_DATA_TYPE_INT64, &t);
    g_ort->ReleaseMemoryInfo(mi);
    if (st) { fprintf(stderr, "Error: Failed to create tensor: %s\n", g_ort->GetErrorMessage(st)); g_ort->ReleaseStatus(st); return NULL; }
    return t;
}

/* /root/reference/src/model.c:81-108.  Unlike the reference (which leaks the flattened buffers on the
 * success path), the tensors made here own their buffer: ReleaseValue frees it. */
int prepare_input_tensors(TokenizedInputs* tok, OrtValue** ids_t, OrtValue** mask_t) {
    if (!tok || !ids_t || !mask_t) return -1;
    int64_t* ids = flatten_int_array(tok->input_ids, tok->batch_size, tok->seq_length);
    if (!ids) return -1;
    *ids_t = create_tensor(ids, tok->batch_size, tok->seq_length);
    if (!*ids_t) { free(ids); return -1; }
    (*ids_t)->owns_data = 1;
    int64_t* mask = flatten_int_array(tok->attention_mask, tok->batch_size, tok->seq_length);
    if (!mask) { g_ort->ReleaseValue(*ids_t); *ids_t = NULL; return -1; }
    *mask_t = create_tensor(mask, tok->batch_size, tok->seq_length);
    if (!*mask_t) { free(mask); g_ort->ReleaseValue(*ids_t); *ids_t = NULL; return -1; }
    (*mask_t)->owns_data = 1;
    return 0;
}

void initialize_ort_api() { g_ort = OrtGetApiBase()->GetApi(ORT_API_VERSION); } /* /root/reference/src/model.c:303-305 */

OrtEnv* initialize_ort_environment() { /* /root/reference/src/model.c:288-298 */
    if (!g_ort) { fprintf(stderr, "Error: initialize_ort_api() was not called\n"); return NULL; }
    OrtEnv* env = NULL;
    OrtStatus* st = g_ort->CreateEnv(ORT_LOGGING_LEVEL_WARNING, "GLiClass", &env);
    if (st) { fprintf(stderr, "Error: Failed to create env for ONNX Runtime: %s\n", g_ort->GetErrorMessage(st)); g_ort->ReleaseStatus(st); return NULL; }
    return env;
}

static int parse_dtype(void) {
    const char* s = getenv("GLICLASS_DTYPE");
    if (!s || !*s || !strcmp(s, "f32") || !strcmp(s, "fp32")) return GLC_F32;
    if (!strcmp(s, "f16") || !strcmp(s, "fp16")) return GLC_F16;
    if (!strcmp(s, "bf16")) return GLC_BF16;
    fprintf(stderr, "Warning: unknown GLICLASS_DTYPE '%s', using f32\n", s);
    return GLC_F32;
}

/* /root/reference/src/model.c:217-281.  num_threads only mattered to ONNXRuntime's CPU thread pools. */
OrtSession* create_ort_session(OrtEnv* env, const char* model_path, int num_threads) {
    (void)num_threads;
    if (!env || !model_path) { fprintf(stderr, "Error: Failed to create session: null argument\n"); return NULL; }
    const int ndev = glc_device_count();
    if (ndev <= 0) { fprintf(stderr, "Error: Failed to create session: no MI355X/HIP device visible (no CPU path exists)\n"); return NULL; }
    int devs[GLC_MAX_DEVICES], nd = 0;
    const char* ds = getenv("GLICLASS_DEVICES");
    if (!ds || !*ds) devs[nd++] = 0;
    else if (!strcmp(ds, "all")) { for (int i = 0; i < ndev && nd < GLC_MAX_DEVICES; ++i) devs[nd++] = i; }
    else {
        const char* p = ds;
        while (*p && nd < GLC_MAX_DEVICES) {
            char* end;
            long v = strtol(p, &end, 10);
            if (end == p || v < 0 || v >= ndev) { fprintf(stderr, "Error: Failed to create session: bad GLICLASS_DEVICES '%s' (%d visible)\n", ds, ndev); return NULL; }
            devs[nd++] = (int)v;
            p = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') { fprintf(stderr, "Error: Failed to create session: bad GLICLASS_DEVICES '%s'\n", ds); return NULL; }
        }
    }
    glc_weights w;
    if (glc_weights_load(model_path, &w) != 0) { fprintf(stderr, "Error: Failed to create session: cannot load '%s'\n", model_path); return NULL; }
    OrtSession* s = (OrtSession*)calloc(1, sizeof(OrtSession));
    if (!s) { glc_weights_free(&w); return NULL; }
    s->cfg = w.cfg;
    const int dtype = parse_dtype();
    for (int i = 0; i < nd; ++i) {
        glc_engine* e = glc_engine_create(&w.cfg, w.tensors, w.n_tensors, devs[i], dtype);
        if (!e) {
            fprintf(stderr, "Error: Failed to create session: %s\n", glc_last_error());
            for (int k = 0; k < s->n_engines; ++k) glc_engine_destroy(s->engines[k]);
            free(s);
            glc_weights_free(&w);
            return NULL;
        }
        s->engines[s->n_engines] = e;
        pthread_mutex_init(&s->q[s->n_engines].mu, NULL);
        pthread_cond_init(&s->q[s->n_engines].cv, NULL);
        s->devices[s->n_engines++] = devs[i];
    }
    {
        /* Default: on (64 rows) in the fp32 mode, where a row's result does not depend on the batch it rides in (to 1e-6);
         * off in the 16-bit modes, where the merged batch picks other GEMM tile kernels and the f16 / bf16 rounding of a row would
         * then depend on which other calls happened to be queued (opt in with GLICLASS_COALESCE_ROWS). */
        const char* cr = getenv("GLICLASS_COALESCE_ROWS");
        s->coalesce_rows = (cr && *cr) ? atoi(cr) : (dtype == GLC_F32 ? 64 : 0);
    }
    glc_weights_free(&w);
    printf("\tUsing MI355X HIP engine on %d GPU(s).\n", s->n_engines);
    return s;
}

/*
 * Request coalescing.  The reference calls run_inference from an OpenMP team, one batch of BATCH_SIZE = 8 texts per call
 * (/root/reference/main.c:141-149, include/configs.h:4); a forward of 8 rows leaves most of an MI355X idle.  Calls that arrive
 * while the engine is busy queue up; whichever caller finds the engine free becomes the leader and serves the queue: it takes
 * waiting calls in arrival order while their rows fit `coalesce_rows`, pads them to the longest sequence among them (pad id,
 * mask 0 — exactly what pad-to-longest does inside one batch, /root/reference/src/tokenizer.c:77-81), runs ONE forward and
 * hands every caller its own rows.  Rows are independent end to end, so each caller gets what it would have got alone — to
 * 1e-6 in the fp32 mode (tests/test_cli.py compares printed scores), and to the operand-rounding noise of the 16-bit modes
 * (a forward of 8 rows and one of 64 use different GEMM tile kernels, i.e. different fp32 summation orders before the f16 rounding).
 * GLICLASS_COALESCE_ROWS (default 64; 0 = off).
 */
static void run_group(OrtSession* s, glc_engine* e, glc_req** g, int n) {
    int ok = 1;
    if (n == 1) {
        int c_out = 0;
        ok = glc_engine_forward(e, g[0]->ids, g[0]->mask, g[0]->B, g[0]->S, g[0]->logits, g[0]->C, &c_out) == 0;
    } else {
        int Bt = 0, Sm = 0, Cm = 0;
        for (int i = 0; i < n; ++i) { Bt += g[i]->B; if (g[i]->S > Sm) Sm = g[i]->S; if (g[i]->C > Cm) Cm = g[i]->C; }
        int64_t* ids = (int64_t*)malloc((size_t)Bt * Sm * sizeof(int64_t));
        int64_t* mask = (int64_t*)calloc((size_t)Bt * Sm, sizeof(int64_t));
        float* lg = (float*)calloc((size_t)Bt * (Cm ? Cm : 1), sizeof(float));
        ok = ids && mask && lg;
        if (ok) {
            int row = 0;
            for (int i = 0; i < n; ++i)
                for (int b = 0; b < g[i]->B; ++b, ++row) {
                    int64_t* di = ids + (size_t)row * Sm;
                    memcpy(di, g[i]->ids + (size_t)b * g[i]->S, (size_t)g[i]->S * sizeof(int64_t));
                    for (int t = g[i]->S; t < Sm; ++t) di[t] = s->cfg.pad_id;
                    memcpy(mask + (size_t)row * Sm, g[i]->mask + (size_t)b * g[i]->S, (size_t)g[i]->S * sizeof(int64_t));
                }
            int c_out = 0;
            ok = glc_engine_forward(e, ids, mask, Bt, Sm, lg, Cm, &c_out) == 0;
            row = 0;
            for (int i = 0; i < n && ok; ++i)
                for (int b = 0; b < g[i]->B; ++b, ++row)
                    for (int j = 0; j < g[i]->C; ++j) g[i]->logits[(size_t)b * g[i]->C + j] = lg[(size_t)row * Cm + j];
        } else fprintf(stderr, "Error during inference: out of memory\n");
        free(ids); free(mask); free(lg);
    }
    if (!ok && n > 1) {
        /* one bad request (e.g. a class token under mask 0) must not fail its neighbours: serve the members one by one */
        for (int i = 0; i < n; ++i) {
            int c_out = 0;
            const int oki = glc_engine_forward(e, g[i]->ids, g[i]->mask, g[i]->B, g[i]->S, g[i]->logits, g[i]->C, &c_out) == 0;
            if (!oki) fprintf(stderr, "Error during inference: %s\n", glc_last_error());
            g[i]->status = oki ? 1 : -1;
        }
        return;
    }
    if (!ok) fprintf(stderr, "Error during inference: %s\n", glc_last_error());
    for (int i = 0; i < n; ++i) g[i]->status = ok ? 1 : -1;     /* published under the queue lock by the caller */
}

static void serve_coalesced(OrtSession* s, int qi, glc_req* mine) {
    glc_queue* q = &s->q[qi];
    pthread_mutex_lock(&q->mu);
    if (q->tail) q->tail->next = mine; else q->head = mine;
    q->tail = mine;
    if (q->busy) {                                   /* a leader is serving this engine: wait until it has done my call */
        while (mine->status == 0 && q->busy) pthread_cond_wait(&q->cv, &q->mu);
        if (mine->status != 0) { pthread_mutex_unlock(&q->mu); return; }
        /* the leader left before reaching my call (cannot happen: it leaves only on an empty queue) — fall through and lead */
    }
    q->busy = 1;
    while (q->head) {
        glc_req* g[64];
        int n = 0, rows = 0;
        while (q->head && n < 64 && (n == 0 || rows + q->head->B <= s->coalesce_rows)) {
            g[n++] = q->head; rows += q->head->B;
            q->head = q->head->next;
        }
        if (!q->head) q->tail = NULL;
        pthread_mutex_unlock(&q->mu);
        int status[64];
        glc_req tmp[64];
        for (int i = 0; i < n; ++i) tmp[i] = *g[i];              /* work on copies: a caller's struct is only written under the lock */
        {
            glc_req* gp[64];
            for (int i = 0; i < n; ++i) gp[i] = &tmp[i];
            run_group(s, s->engines[qi], gp, n);
            for (int i = 0; i < n; ++i) status[i] = tmp[i].status;
        }
        pthread_mutex_lock(&q->mu);
        for (int i = 0; i < n; ++i) g[i]->status = status[i];
        pthread_cond_broadcast(&q->cv);
    }
    q->busy = 0;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
}

static OrtValue* run_on_engine(OrtSession* s, glc_engine* e, OrtValue* ids_t, OrtValue* mask_t) {
    if (!ids_t || !mask_t || ids_t->type != ONNX_TENSOR_ELEMENT_DATA_TYPE_INT64 || mask_t->type != ONNX_TENSOR_ELEMENT_DATA_TYPE_INT64 ||
        ids_t->ndim != 2 || mask_t->ndim != 2 || ids_t->dims[0] != mask_t->dims[0] || ids_t->dims[1] != mask_t->dims[1]) {
        fprintf(stderr, "Error during inference: input_ids / attention_mask must be INT64 [batch, seq] tensors of equal shape\n");
        return NULL;
    }
    const int B = (int)ids_t->dims[0], S = (int)ids_t->dims[1];
    const int64_t* ids = (const int64_t*)ids_t->data;
    /* the graph's dynamic output width: max number of <<LABEL>> tokens in a row */
    int C = 0;
    for (int b = 0; b < B; ++b) {
        int c = 0;
        for (int t = 0; t < S; ++t) c += ids[(size_t)b * S + t] == s->cfg.class_token_index;
        if (c > C) C = c;
    }
    float* logits = (float*)calloc((size_t)B * (C ? C : 1), sizeof(float));
    if (!logits) { fprintf(stderr, "Error during inference: out of memory\n"); return NULL; }
    int qi = 0;
    while (qi < s->n_engines && s->engines[qi] != e) ++qi;
    if (s->coalesce_rows > 1 && qi < s->n_engines) {
        glc_req r = {ids, (const int64_t*)mask_t->data, B, S, C, logits, 0, NULL};
        serve_coalesced(s, qi, &r);
        if (r.status != 1) { free(logits); return NULL; }
    } else {
        int c_out = 0;
        if (glc_engine_forward(e, ids, (const int64_t*)mask_t->data, B, S, logits, C, &c_out) != 0) {
            fprintf(stderr, "Error during inference: %s\n", glc_last_error());
            free(logits);
            return NULL;
        }
    }
    int64_t dims[2] = {B, C};
    OrtValue* out = glc_value_new(ONNX_TENSOR_ELEMENT_DATA_TYPE_FLOAT, dims, 2, logits, 1);
    if (!out) free(logits);
    return out; /* caller releases with g_ort->ReleaseValue (/root/reference/src/model.c:204-206) */
}

/* /root/reference/src/model.c:122-207.  Safe to call concurrently on one session
 * (/root/reference/main.c:141-149): calls are dealt round-robin to the session's GPUs and each engine
 * serialises internally. */
OrtValue* run_inference(OrtSession* session, OrtValue* ids_t, OrtValue* mask_t) {
    if (!session || session->n_engines <= 0) { fprintf(stderr, "Error during inference: invalid session\n"); return NULL; }
    unsigned k = __atomic_fetch_add(&session->next, 1u, __ATOMIC_RELAXED) % (unsigned)session->n_engines;
    return run_on_engine(session, session->engines[k], ids_t, mask_t);
}

void parallel_inference(OrtSession* session, OrtValue** ids_ts, OrtValue** mask_ts, size_t num_batches, OrtValue** outs) {
    if (!session || session->n_engines <= 0) { for (size_t i = 0; i < num_batches; ++i) outs[i] = NULL; return; }
    const int G = session->n_engines;
    for (size_t i = 0; i < num_batches; ++i) outs[i] = NULL;
    /* batch i belongs to engine i % G; the team may be smaller than G (nested region, OMP_THREAD_LIMIT): stride by its real size
     * so that every batch is served whatever the runtime grants */
#pragma omp parallel num_threads(G)
    {
        const size_t t = (size_t)omp_get_thread_num(), nt = (size_t)omp_get_num_threads();
        for (size_t i = t; i < num_batches; i += nt) outs[i] = run_on_engine(session, session->engines[i % (size_t)G], ids_ts[i], mask_ts[i]);
    }
}

int glc_session_num_devices(const OrtSession* session) { return session ? session->n_engines : 0; }
