/* A stand-in for avdsp_kernels.hip behind include/avdsp_hip.h, for ONE purpose: running the C host runtime (avdsp_host.c) under
 * AddressSanitizer on a machine without a GPU (tests/dev/host_asan/run.sh).  Development aid, not product and not oracle: it computes
 * nothing.  "Device" memory is exact-size heap memory, so that every range the host hands over is touched and a range that is too long
 * for the buffer behind it is an ASan report; block calls read the whole input window, rewrite the whole output window with what it
 * held, and return.  Tests that compare samples therefore fail under it -- run.sh looks for sanitizer reports only. */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "avdsp_hip.h"

#define MAXPLANS 4096
struct avdsp_hip_prog {
    int total_words, chain_inst, inst_n;
    int32_t *buf; size_t buf_words;
    int nplans; int generic[MAXPLANS]; int strands[MAXPLANS]; int instances[MAXPLANS];
    int opts[32];
};
static __thread char g_err[256];
static volatile unsigned g_sink;

static int fail(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}
static void touch_r(const void *p, size_t bytes) { const unsigned char *c = p; unsigned s = 0; for (size_t i = 0; i < bytes; i++) s += c[i]; g_sink += s; }
static void touch_rw(void *p, size_t bytes) { unsigned char *c = p; for (size_t i = 0; i < bytes; i++) { unsigned char v = c[i]; c[i] = v; } }

int avdsp_hip_device_count(void) { return 1; }
int avdsp_hip_set_device(int o) { return o == 0 ? 0 : fail("no device %d", o); }
const char *avdsp_hip_last_error(void) { return g_err; }
int avdsp_hip_last_error_is_ready_timeout(void) { return 0; }
int avdsp_hip_synchronize(void *s) { (void)s; return 0; }

avdsp_hip_prog *avdsp_hip_prog_create(int total_words)
{
    if (total_words <= 0) { fail("empty program"); return NULL; }
    avdsp_hip_prog *p = calloc(1, sizeof *p);
    p->total_words = total_words; p->buf_words = (size_t)total_words; p->buf = calloc(p->buf_words, 4); p->inst_n = 1;
    p->opts[AVDSP_OPT_GROUP_FANOUT] = 1; p->opts[AVDSP_OPT_SIDE_BY_SIDE] = -1;
    return p;
}
void avdsp_hip_prog_destroy(avdsp_hip_prog *p) { if (p) { free(p->buf); free(p); } }

static int new_plan(avdsp_hip_prog *p, int generic, int instances)
{
    if (p->nplans >= MAXPLANS) return fail("stub: too many plans");
    p->generic[p->nplans] = generic; p->strands[p->nplans] = 0; p->instances[p->nplans] = instances;
    return p->nplans++;
}

int avdsp_hip_prog_add_plan(avdsp_hip_prog *p, const avdsp_plan_desc *d)
{
    if (d->format < 2 || d->format > 6) return fail("format %d", d->format);
    const long long words = d->instances > 1 ? (long long)AVDSP_INSTANCE_STRIDE(p->total_words) * d->instances : p->total_words;
    if (d->instances > 1 && d->instances != p->chain_inst) return fail("a plan of %d instances, the device holds %d copies", d->instances, p->chain_inst);
    touch_r(d->chains, (size_t)d->nchains * sizeof *d->chains);
    touch_r(d->sec_coef_word, (size_t)d->nsections * 4); touch_r(d->sec_state_word, (size_t)d->nsections * 4);
    for (int i = 0; i < d->nsections; i++) {
        const long long c = d->sec_coef_word[i], s = d->sec_state_word[i];
        if (c < 0 || c + 5 > words || s < 0 || s + 6 > words || (s & 1)) return fail("section %d addresses words outside the loaded buffer", i);
        touch_r(p->buf + c, 20); touch_rw(p->buf + s, 24);
    }
    for (int i = 0; i < d->nchains; i++) {
        const avdsp_chain *c = &d->chains[i];
        if (c->sec_base < 0 || c->nsec < 0 || c->sec_base + c->nsec > d->nsections) return fail("chain %d: bad section range", i);
        if (c->n_out < 1 || c->n_out > AVDSP_MAX_STORES || c->in_io < 0) return fail("chain %d: bad IO", i);
        if (c->fir_taps) {
            if (c->fir_coef_word < 0 || c->fir_coef_word + (long long)c->fir_taps > words || c->fir_state_word < 0 || c->fir_state_word + (long long)c->fir_taps > words)
                return fail("chain %d: FIR addresses words outside the loaded buffer", i);
            touch_r(p->buf + c->fir_coef_word, (size_t)c->fir_taps * 4); touch_rw(p->buf + c->fir_state_word, (size_t)c->fir_taps * 4);
        }
    }
    return new_plan(p, 0, d->instances > 1 ? d->instances : 1);
}

int avdsp_hip_prog_add_generic(avdsp_hip_prog *p, const avdsp_generic_desc *d)
{
    if (d->nown > 0) {
        touch_r(d->own, (size_t)d->nown * 8);
        for (int i = 0; i < d->nown; i++) {
            const long long lo = d->own[2 * i], hi = d->own[2 * i + 1];
            if (lo < 0 || hi < lo || hi > p->total_words + 2) return fail("own range %d: [%lld, %lld) outside the mirror", i, lo, hi);
        }
    }
    if (d->core_word < 0 || d->core_word >= p->total_words || d->end_word < 0 || d->end_word > p->total_words) return fail("core words outside the mirror");
    for (int i = 0; i < d->nvm; i++) if (d->vm_word[i] < 0 || d->vm_word[i] >= p->total_words) return fail("vm word outside the mirror");
    return new_plan(p, 1, 1);
}
int avdsp_hip_plan_add_strands(avdsp_hip_prog *p, int plan, const avdsp_strand_desc *d)
{
    if (plan < 0 || plan >= p->nplans) return fail("bad plan id %d", plan);
    touch_r(d->ops, (size_t)d->nops * sizeof *d->ops);
    touch_r(d->args, (size_t)d->nstrands * d->nargs * 4);
    for (int i = 0; i < d->nops; i++)
        for (int k = 0; k < 3; k++) {
            const int a = k == 0 ? d->ops[i].a0 : k == 1 ? d->ops[i].a1 : d->ops[i].a2;
            if (a < -1 || a >= d->nargs) return fail("strand op %d: argument column %d of %d", i, a, d->nargs);
        }
    p->strands[plan] = d->nstrands;
    return 0;
}
int avdsp_hip_plan_strands(const avdsp_hip_prog *p, int plan) { return plan >= 0 && plan < p->nplans ? p->strands[plan] : 0; }
int avdsp_hip_tpdf_reset(avdsp_hip_prog *p, int seed, int dd) { (void)p; (void)seed; (void)dd; return 0; }
int avdsp_hip_prog_clear_plans(avdsp_hip_prog *p) { p->nplans = 0; return 0; }

static int check_range(avdsp_hip_prog *p, int first, int n)
{
    if (first < 0 || n < 0 || (long long)first + n > p->total_words) return fail("words [%d, %d + %d) outside the mirror of %d", first, first, n, p->total_words);
    return 0;
}
int avdsp_hip_upload_words(avdsp_hip_prog *p, const int32_t *h, int first, int n)
{
    if (check_range(p, first, n)) return -1;
    memcpy(p->buf + first, h + first, (size_t)n * 4); return 0;
}
int avdsp_hip_download_words(avdsp_hip_prog *p, int32_t *h, int first, int n)
{
    if (check_range(p, first, n)) return -1;
    memcpy(h + first, p->buf + first, (size_t)n * 4); return 0;
}
int avdsp_hip_zero_words(avdsp_hip_prog *p, int first, int n)
{
    if (check_range(p, first, n)) return -1;
    memset(p->buf + first, 0, (size_t)n * 4); return 0;
}
int avdsp_hip_chain_instances(avdsp_hip_prog *p, int n)
{
    if (n < 0 || n > 65536) return fail("instances: 1 .. 65536");
    if (p->nplans) return fail("chain instances: the program's plans must be dropped first");
    if (n <= 1 && p->chain_inst <= 1) { p->chain_inst = 0; return 0; }
    const size_t W = (size_t)p->total_words, Wp = AVDSP_INSTANCE_STRIDE(W), copies = n > 1 ? (size_t)n : 1;
    int32_t *nb = calloc(copies * Wp, 4);
    for (size_t i = 0; i < copies; i++) memcpy(nb + i * Wp, p->buf, W * 4);
    free(p->buf); p->buf = nb; p->buf_words = copies * Wp; p->chain_inst = n > 1 ? n : 0;
    return 0;
}
int avdsp_hip_set_instances(avdsp_hip_prog *p, int n) { if (n < 1 || n > 65536) return fail("instances: 1 .. 65536"); p->inst_n = n; return 0; }
int avdsp_hip_download_instance_words(avdsp_hip_prog *p, int inst, int32_t *h, int first, int n)
{
    const int have = p->inst_n > p->chain_inst ? p->inst_n : p->chain_inst;
    if (inst < 0 || inst >= have) return fail("instance %d of %d", inst, have);
    if (check_range(p, first, n)) return -1;
    if (p->chain_inst > 1) memcpy(h, p->buf + (size_t)inst * AVDSP_INSTANCE_STRIDE(p->total_words) + first, (size_t)n * 4);
    else memcpy(h, p->buf + first, (size_t)n * 4);
    return 0;
}

/* block calls over HOST buffers: every word of both windows is touched (device-pointer calls cannot be: there is no device) */
static int block_host(avdsp_hip_prog *p, int plan, const void *in, int in_stride, size_t in_elem, void *out, int out_stride, int nframes)
{
    if (plan < 0 || plan >= p->nplans) return fail("bad plan id %d", plan);
    if (nframes <= 0) return 0;
    if (in_stride > 0) touch_r(in, (size_t)nframes * in_stride * in_elem);
    if (out_stride > 0) touch_rw(out, (size_t)nframes * out_stride * 4);
    return 0;
}
int avdsp_hip_run_block_host(avdsp_hip_prog *p, int plan, const void *h_in, int is, int ib, void *h_out, int os, int ob, int nf, int fi, int bi)
{
    (void)ib; (void)ob; (void)fi; (void)bi;
    return block_host(p, plan, h_in, is, 4, h_out, os, nf);
}
int avdsp_hip_submit_block_host(avdsp_hip_prog *p, int plan, const void *h_in, int is, int ib, void *h_out, int os, int ob, int nf, int fi, int bi)
{
    (void)ib; (void)ob; (void)fi; (void)bi;
    return block_host(p, plan, h_in, is, 4, h_out, os, nf);
}
int avdsp_hip_wait_block_host(avdsp_hip_prog *p, int m) { (void)p; (void)m; return 0; }
static size_t pcm_bytes(int pcm) { return pcm == AVDSP_PCM_S32 ? 4 : pcm == AVDSP_PCM_S24_3LE ? 3 : 2; }
int avdsp_hip_run_block_pcm_host(avdsp_hip_prog *p, int plan, int pcm, const void *h_src, int is, int ib, void *h_out, int os, int ob, int nf, int fi, int bi)
{
    (void)ib; (void)ob; (void)fi; (void)bi;
    return block_host(p, plan, h_src, is, pcm_bytes(pcm), h_out, os, nf);
}
static int levels_ok(avdsp_hip_prog *p, const int *plans, const int *level_size, int nlevels)
{
    int k = 0;
    for (int l = 0; l < nlevels; l++)
        for (int i = 0; i < level_size[l]; i++, k++)
            if (plans[k] < 0 || plans[k] >= p->nplans) return fail("bad plan id %d", plans[k]);
    return k ? 0 : 0;
}
int avdsp_hip_run_levels_host(avdsp_hip_prog *p, const int *plans, const int *ls, int nl, const void *h_in, int is, int ib, void *h_out, int os, int ob, int nf, int fi, int bi)
{
    (void)ib; (void)ob; (void)fi; (void)bi;
    if (levels_ok(p, plans, ls, nl)) return -1;
    return nl ? block_host(p, plans[0], h_in, is, 4, h_out, os, nf) : 0;
}
int avdsp_hip_run_levels_pcm_host(avdsp_hip_prog *p, const int *plans, const int *ls, int nl, int pcm, const void *h_src, int is, int ib, void *h_out, int os, int ob, int nf, int fi, int bi)
{
    (void)ib; (void)ob; (void)fi; (void)bi;
    if (levels_ok(p, plans, ls, nl)) return -1;
    return nl ? block_host(p, plans[0], h_src, is, pcm_bytes(pcm), h_out, os, nf) : 0;
}
/* device-pointer calls: plan ids only */
int avdsp_hip_run_block(avdsp_hip_prog *p, int plan, const void *di, int is, int ib, void *dout, int os, int ob, int nf, int fi, int bi, void *st)
{
    (void)di; (void)is; (void)ib; (void)dout; (void)os; (void)ob; (void)nf; (void)fi; (void)bi; (void)st;
    return plan >= 0 && plan < p->nplans ? 0 : fail("bad plan id %d", plan);
}
int avdsp_hip_run_levels(avdsp_hip_prog *p, const int *plans, const int *ls, int nl, const void *di, int is, int ib, void *dout, int os, int ob, int nf, int fi, int bi, void *st)
{
    (void)di; (void)is; (void)ib; (void)dout; (void)os; (void)ob; (void)nf; (void)fi; (void)bi; (void)st;
    return levels_ok(p, plans, ls, nl);
}
int avdsp_hip_run_levels_instances(avdsp_hip_prog *p, const int *plans, const int *ls, int nl, const void *di, int is, int ib, size_t iw,
                                   void *dout, int os, int ob, size_t ow, int nf, void *st)
{
    (void)di; (void)is; (void)ib; (void)iw; (void)dout; (void)os; (void)ob; (void)ow; (void)nf; (void)st;
    return levels_ok(p, plans, ls, nl);
}
int avdsp_hip_unpack_pcm(avdsp_hip_prog *p, int pcm, const void *s, void *d, size_t n, void *st) { (void)p; (void)pcm; (void)s; (void)d; (void)n; (void)st; return 0; }
int avdsp_hip_profile_enable(avdsp_hip_prog *p, int on) { (void)p; (void)on; return 0; }
int avdsp_hip_profile_read(avdsp_hip_prog *p, int kind, double *ms, int *launches) { (void)p; (void)kind; if (ms) *ms = 0; if (launches) *launches = 0; return 0; }
int avdsp_hip_profile_last_pairs(avdsp_hip_prog *p, int kind) { (void)p; (void)kind; return 0; }
int avdsp_hip_prog_get_option(avdsp_hip_prog *p, int key) { return key >= 0 && key < 32 ? p->opts[key] : -1; }
int avdsp_hip_prog_set_option(avdsp_hip_prog *p, int key, int value) { if (key < 0 || key >= 32) return fail("unknown option %d", key); p->opts[key] = value; return 0; }
int avdsp_hip_ready_clear(avdsp_hip_prog *p) { (void)p; return 0; }
int avdsp_hip_ready_timeouts(avdsp_hip_prog *p) { (void)p; return 0; }
int avdsp_hip_tag_output(avdsp_hip_prog *p, void *c, int stride, int nf, int reset, int rv, void *st) { (void)p; (void)c; (void)stride; (void)nf; (void)reset; (void)rv; (void)st; return 0; }
int avdsp_hip_tag_column_host(avdsp_hip_prog *p, int *h, int nf) { (void)p; touch_rw(h, (size_t)nf * 4); return 0; }
