/*
 * avdsp_host.c -- host side of the MI355X-native AVDSP runtime, plain C99.
 *
 * Implements the reference's C API (module_avdsp/runtime/dsp_runtime.h:160-164) on top of the thin
 * HIP ABI in include/avdsp_hip.h.  This file owns: program validation (dspRuntimeInit,
 * dsp_runtime.c:150-195), sample-rate selection and state reset (dspRuntimeReset, :116-145), core
 * lookup (:42-77) and the *lowering* of a core's opcode stream into channel chains for the device.
 * It contains no signal arithmetic: every sample is computed by the gfx950 kernels.  A core that
 * is a set of independent  LOAD|LOAD_GAIN -> BIQUADS* -> [FIR] -> [SAT0DB] -> STORE+  chains goes to
 * the parallel kernels; any other core is bounds-checked here (scan_generic) and handed to the
 * general device interpreter.  There is no CPU fallback.
 */
#include "avdsp_runtime.h"
#include "avdsp_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- exported data of the reference runtime (dsp_runtime.c:36-38) ---- */
dspHeader_t *dspHeaderPtr = 0;
int          dspBiquadFreqSkip = 0;
int          dspMantissa = DSP_MANT;

/* dsp_header.c:10-73 (the leading newlines on four entries are the reference's) */
const char *dspOpcodeText[DSP_MAX_OPCODE] = {
    "DSP_END_OF_CODE", "\nDSP_HEADER", "DSP_NOP", "\nDSP_CORE", "\nDSP_PARAM", "\nDSP_PARAM_NUM",
    "DSP_SERIAL", "DSP_TPDF_CALC", "DSP_TPDF", "DSP_WHITE", "DSP_CLRXY", "DSP_SWAPXY", "DSP_COPYXY",
    "DSP_COPYYX", "DSP_ADDXY", "DSP_ADDYX", "DSP_SUBXY", "DSP_SUBYX", "DSP_MULXY", "DSP_DIVXY",
    "DSP_DIVYX", "DSP_AVGXY", "DSP_AVGYX", "DSP_NEGX", "DSP_NEGY", "DSP_SQRTX", "DSP_SHIFT",
    "DSP_VALUE", "DSP_VALUE_INT", "DSP_MUL_VALUE", "DSP_MUL_VALUE_INT", "DSP_DIV_VALUE",
    "DSP_DIV_VALUE_INT", "DSP_AND_VALUE_INT", "DSP_LOAD", "DSP_LOAD_GAIN", "DSP_LOAD_MUX",
    "DSP_STORE", "DSP_LOAD_STORE", "DSP_LOAD_MEM", "DSP_STORE_MEM", "DSP_GAIN", "DSP_SAT0DB",
    "DSP_SAT0DB_TPDF", "DSP_SAT0DB_GAIN", "DSP_SAT0DB_TPDF_GAIN", "DSP_DELAY_1", "DSP_DELAY",
    "DSP_DELAY_DP", "DSP_DATA_TABLE", "DSP_BIQUADS", "DSP_FIR", "DSP_RMS", "DSP_DCBLOCK",
    "DSP_DITHER", "DSP_DITHER_NS2", "DSP_DISTRIB", "DSP_DIRAC", "DSP_SQUAREWAVE", "DSP_CLIP",
    "DSP_LOAD_MEM_DATA", "DSP_SINE"
};

/* dspQNM / dspQM64 / dspQM32 (dsp_header.c:75-85) live in avdsp_qformat.c, shared with the encoder library */

/* ---- the one loaded program (the reference keeps the same things in file-scope statics) ---- */
/* What a core reads and writes outside its opcode-private use of the frame: decides which cores of a program may
 * run side by side (dspRuntimeBlockAll) and what an interpreter launch writes back.  Filled by scan_generic. */
#define DEPS_MAX_RANGES 128
typedef struct {
    int      complete;                   /* 0: more than these tables hold, or a chain core: runs alone */
    unsigned early_io[8], written_io[8]; /* frame slots (< 256) read before written in a frame / stored */
    int      nwr, wr_word[64];           /* mirror words written as per-frame variables (STORE_MEM, result words) */
    int      nrd, rd_word[64];           /* mirror words read before this core writes them in the frame */
    int      nranges, range[DEPS_MAX_RANGES][2];   /* opcode-private state, mirror words [lo, hi) */
    int      nparams, param[DEPS_MAX_RANGES][2];   /* program words read as parameters, merged, [lo, hi) */
    int      tpdf_calc, tpdf_user;
} core_deps;

typedef struct {
    opcode_t   *core;            /* key: the pointer the host passes (after dspFindCoreBegin) */
    int         format;
    int         end_word;        /* 0: the whole core; else a strand group [core, end_word) of it */
    int         skip_from;       /* ... that leaves [skip_from, skip_to) out (0: nothing) */
    int         plan_id;         /* id inside the device program, < 0 = lowering failed */
    int         nchains, max_sections, max_taps;      /* nchains 0 = general interpreter */
    int         total_chains, first_chain;            /* chain cores: the core has total_chains, this process runs
                                                         [first_chain, first_chain + nchains) of them (dspRuntimeSetShard) */
    int         empty;                                /* chain core whose shard holds no chain: block calls do nothing */
    core_deps   deps;
} core_plan;

#define MAX_CORE_PLANS 512          /* cores, and the strand groups dspRuntimeBlockAll cuts them into */
#define MAX_ARRANGEMENTS 17
typedef struct { int valid, format, n, nlevels, ncores; opcode_t *only; int *plans, *size; } arrangement;

/* One loaded program = one context, found again by any pointer into its buffer (dspRuntimeInit's codePtr owns it, as the
 * reference's host owns "opcodes").  The reference keeps this in file statics, one program per process; here the statics
 * of the program the latest call named are swapped in (dspHeaderPtr, dspBiquadFreqSkip, dspMantissa follow), so several
 * programs -- on several GPUs -- live in one process.  Calls that name no program (dspRuntimeReset, options, shard,
 * timers, tagoutput, wait) address the program of the latest call that did, or dspRuntimeSelect's. */
typedef struct avdsp_ctx {
    opcode_t       *code;
    int             total_words;         /* totalLength + dataSize */
    int             have_rate;
    int             freq_index, num_freq, biquad_offset;
    int             dither, store_mask, random, fs_index;
    avdsp_hip_prog *dev;
    int             dev_state_valid;     /* device mirror holds the authoritative state */
    core_plan       plans[MAX_CORE_PLANS];
    int             nplans;
    int             opt_fir_impl, opt_biquad_impl, opt_device, opt_profile, opt_generic, opt_interp_impl;
    int             device_selected;
    int             last_levels, last_cores, last_pieces;  /* of the latest dspRuntimeBlockAll: dspRuntimeGetOption("levels" / "cores" / "pieces") */
    int             opt_strand_split, next_tpdf_role;
    int             next_skip_from, next_skip_to;          /* the piece being scanned / planned leaves [from, to) out (0: nothing) */
    int             opt_strand_lanes;                      /* uniform strand runs of an interpreted core on lanes (strand_lower) */
    const avdsp_strand_desc *next_strands;                 /* get_plan_range attaches this to the plan it makes */
    int             last_strands;                          /* strands of the latest dspRuntimeBlockAll arrangement that run on lanes */
    int             shard_rank, shard_world;               /* dspRuntimeSetShard: this process's slice of every chain core */
    int             numfreq_at_init;                       /* dspChangeFormat's view of the rate count (see dspRuntimeInit) */
    int             opt_profile_stride;
    int             opt_overlap, opt_fir_rows, opt_host_split, opt_host_pin, opt_ready_words, opt_lane_hw, opt_fir_split, opt_fir_launch, opt_fir_launch_set, opt_fir_lean, opt_fir_lean_set, opt_ring_wait; /* launch arrangement of the chain kernels (avdsp_hip_prog_set_option) */
    arrangement     arr[MAX_ARRANGEMENTS]; int arr_next;   /* how the cores / pieces go to the device: [0] whole program, [1..] single cores */
    int             biquad_freq_skip, mantissa;            /* this program's dspBiquadFreqSkip / dspMantissa */
    int             device_ordinal;                        /* the GPU its device copy lives on (-1: none yet) */
    int             ninst;                                 /* dspRuntimeSetInstances */
    int             inst_saved_lanes;                      /* 1 + the "strand_lanes" the program had before it got instances (0: nothing saved) */
    int             inst_saved_generic;                    /* 1 + the "generic" a program of BOTH kinds of cores had before its instances put every core on the interpreter */
    int             inst_call;                             /* inside dspRuntimeBlockAllInstancesDevice: block_all hands these strides on */
    size_t          inst_in_words, inst_out_words;
    int             opt_cu_split;                          /* experiment: "cu_split" */
    int             opt_group_serial;                      /* "group_fanout" 0 (stored inverted: the default, 0, is fan-out on) */
    int             inst_chain_mode;                       /* 0 not looked yet, 1 every core runs on the interpreter, 2 every core is a chain core (round 5) */
    int             chain_inst_made;                       /* the device holds ninst copies of the mirror; the plans are ninst x the cores' chains, made for ... */
    size_t          chain_inst_in, chain_inst_out;         /* ... these distances between the instances' sample blocks (words) */
    int             chain_inst_win[5];                     /* the windows the cores' IOs were last checked against (format, in base, in stride, out base, out stride); [0] = 0: none */
} avdsp_ctx;

/* no program loaded: options set now are the defaults every program starts from (and keeps following, see dspRuntimeSetOption) */
static avdsp_ctx g_template = { .opt_lane_hw = 1, .opt_ring_wait = 1, .opt_ready_words = -1, .opt_fir_impl = 1, .opt_biquad_impl = 1, .opt_device = -1, .opt_interp_impl = 1, .opt_strand_split = 1, .opt_strand_lanes = 1, .shard_world = 1,
                                .mantissa = DSP_MANT, .device_ordinal = -1 };
#define MAX_PROGRAMS 64
static avdsp_ctx *g_ctx[MAX_PROGRAMS];
static int        g_nctx;
static avdsp_ctx *g_cur = &g_template;
#define G (*g_cur)
static int        g_rate_static;     /* the reference's dspNumSamplingFreq static: what the latest dspRuntimeReset of the PROCESS left behind */

static char g_err[512];
static int  g_err_code;
static int  ensure_encoding(int format);

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    g_err_code = code;
    return code;
}

const char *dspRuntimeLastError(void) { return g_err; }

/* a failure of the HIP layer: -10 with its text -- except the sticky one, a FIR wave that gave up waiting for its cascade's ready
 * word in an earlier block (avdsp_kernels.hip, ready_check): -11, from every entry point until dspRuntimeReset() or
 * dspRuntimeSetOption("ready_timeouts", 0) acknowledges it.  (dsp_runtime.c:150-195: the reference's failures are return codes.) */
static int hip_fail(void)
{
    return fail(avdsp_hip_last_error_is_ready_timeout() ? -11 : -10, "%s", avdsp_hip_last_error());
}

static void drop_device(void)
{
    if (G.dev) avdsp_hip_prog_destroy(G.dev);
    G.dev = 0;
    G.nplans = 0;
    for (int i_ = 0; i_ < MAX_ARRANGEMENTS; i_++) G.arr[i_].valid = 0;
    G.dev_state_valid = 0;
    G.chain_inst_made = 0;
    G.chain_inst_win[0] = 0;
    G.inst_chain_mode = 0;               /* (which kind of program it is gets looked at again: another program, another option) */
}

/* swap a program's statics in: the exported globals of the reference runtime follow, and so does the active GPU */
static void ctx_make_current(avdsp_ctx *c)
{
    if (c == g_cur) return;
    if (g_cur != &g_template) { g_cur->biquad_freq_skip = dspBiquadFreqSkip; g_cur->mantissa = dspMantissa; }
    g_cur = c;
    dspHeaderPtr = (dspHeader_t *)c->code;
    dspBiquadFreqSkip = c->biquad_freq_skip;
    dspMantissa = c->mantissa;
}

/* The HIP current device is per thread and anybody's to change (torch in the tests and in bench.py calls hipSetDevice
 * itself): every entry point that may reach the device names the current program's GPU again instead of trusting a
 * remembered one.  The library as a whole is single-threaded, like the reference (statics, no locks): one thread at a
 * time may be inside it. */
static void device_current(void)
{
    if (G.device_selected && G.device_ordinal >= 0) (void)avdsp_hip_set_device(G.device_ordinal);
}

/* words [code, code + total_words) belong to a program (half open: two programs may lie back to back in one array) */
static int ctx_holds(const avdsp_ctx *c, const opcode_t *q)
{
    return c->code && q >= c->code && q < c->code + (c->total_words > 0 ? c->total_words : 1);
}

/* the program a pointer belongs to (its code or its data area); the current one stays if none does */
static avdsp_ctx *ctx_of(const void *ptr)
{
    const opcode_t *q = (const opcode_t *)ptr;
    avdsp_ctx *hit = 0;
    if (g_cur != &g_template && ctx_holds(g_cur, q)) hit = g_cur;
    for (int i = 0; i < g_nctx && !hit; i++)
        if (ctx_holds(g_ctx[i], q)) hit = g_ctx[i];
    /* one past the last word: the rundata pointer of a program without a data area (checked last: it may also be the
     * first word of the next program in the same array) */
    if (!hit && g_cur != &g_template && g_cur->code && q == g_cur->code + g_cur->total_words) hit = g_cur;
    for (int i = 0; i < g_nctx && !hit; i++)
        if (g_ctx[i]->code && q == g_ctx[i]->code + g_ctx[i]->total_words) hit = g_ctx[i];
    if (hit) ctx_make_current(hit);
    device_current();
    return g_cur;
}

static void ctx_destroy(avdsp_ctx *c)
{
    avdsp_ctx *back = g_cur != c ? g_cur : &g_template;      /* whoever was current stays current */
    ctx_make_current(c);
    drop_device();
    for (int i = 0; i < MAX_ARRANGEMENTS; i++) free(c->arr[i].plans);      /* (size[] lives in the same allocation) */
    for (int i = 0; i < g_nctx; i++)
        if (g_ctx[i] == c) { g_ctx[i] = g_ctx[--g_nctx]; break; }
    g_cur = &g_template;
    dspHeaderPtr = 0; dspBiquadFreqSkip = 0; dspMantissa = DSP_MANT;
    free(c);
    if (back != &g_template) ctx_make_current(back);
}

void dspRuntimeRelease(void)
{
    while (g_nctx) ctx_destroy(g_ctx[g_nctx - 1]);
    g_cur = &g_template;
    dspHeaderPtr = 0;
}

/* one program of several: frees its device memory and forgets it (the buffer is the caller's) */
int dspRuntimeReleaseProgram(opcode_t *codePtr)
{
    for (int i = 0; i < g_nctx; i++)
        if (g_ctx[i]->code == codePtr) { ctx_destroy(g_ctx[i]); return 0; }
    return fail(-1, "dspRuntimeReleaseProgram: no program loaded at that address");
}

/* makes the program a pointer belongs to the one that calls without a program pointer address */
int dspRuntimeSelect(const void *ptr_into_program)
{
    avdsp_ctx *c = ctx_of(ptr_into_program);
    const opcode_t *q = (const opcode_t *)ptr_into_program;
    if (c == &g_template || !c->code || !(ctx_holds(c, q) || q == c->code + c->total_words))
        return fail(-1, "dspRuntimeSelect: the pointer is inside no loaded program");
    return 0;
}

/* a tunable that changes how cores are lowered: forget the plans, keep the device copy (state, TPDF globals) */
static int replan(void)
{
    device_current();
    if (G.dev && avdsp_hip_prog_clear_plans(G.dev)) return hip_fail();
    G.nplans = 0;
    for (int i_ = 0; i_ < MAX_ARRANGEMENTS; i_++) G.arr[i_].valid = 0;
    G.chain_inst_win[0] = 0;
    if (!G.chain_inst_made) G.inst_chain_mode = 0;
    return 0;
}

static int set_option_here(const char *key, int value)
{
    device_current();
    if (!strcmp(key, "fir_impl"))    { G.opt_fir_impl = value; return 0; }
    if (!strcmp(key, "biquad_impl")) { G.opt_biquad_impl = value; return 0; }
    if (!strcmp(key, "device"))      { G.opt_device = value; G.device_selected = 0; return 0; }
    if (!strcmp(key, "generic"))     { G.opt_generic = value; return replan(); }
    if (!strcmp(key, "interp_impl")) { G.opt_interp_impl = value; return replan(); }
    if (!strcmp(key, "strand_split")) { G.opt_strand_split = value; return replan(); }
    if (!strcmp(key, "strand_lanes")) { G.opt_strand_lanes = value; return replan(); }
    if (!strcmp(key, "overlap") || !strcmp(key, "fir_rows") || !strcmp(key, "host_split") || !strcmp(key, "ready_words") || !strcmp(key, "lane_hw") || !strcmp(key, "fir_split")) {
        const int split = !strcmp(key, "fir_split");
        int *slot = split ? &G.opt_fir_split : key[0] == 'o' ? &G.opt_overlap : key[0] == 'f' ? &G.opt_fir_rows : key[0] == 'r' ? &G.opt_ready_words : key[0] == 'l' ? &G.opt_lane_hw : &G.opt_host_split;
        const int dev_key = split ? AVDSP_OPT_FIR_SPLIT : key[0] == 'o' ? AVDSP_OPT_OVERLAP : key[0] == 'f' ? AVDSP_OPT_FIR_ROWS : key[0] == 'r' ? AVDSP_OPT_READY_WORDS : key[0] == 'l' ? AVDSP_OPT_LANE_HW : AVDSP_OPT_HOST_SPLIT;
        if (G.dev && avdsp_hip_prog_set_option(G.dev, dev_key, value)) return hip_fail();
        *slot = value;
        return 0;
    }
    /* the reference's dspNumSamplingFreq static as a fresh process would have it (0), or as an earlier program left it:
     * what dspChangeFormat converts depends on it (dspRuntimeInit); a test hook, a process restart does the same */
    if (!strcmp(key, "rate_count_static")) { g_rate_static = value; return 0; }
    if (!strcmp(key, "host_pin")) {
        if (G.dev && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_HOST_PIN, value)) return hip_fail();
        G.opt_host_pin = value;
        return 0;
    }
    if (!strcmp(key, "fir_lean")) {
        if (G.dev && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_FIR_LEAN, value)) return hip_fail();
        G.opt_fir_lean = value; G.opt_fir_lean_set = 1;
        return 0;
    }
    if (!strcmp(key, "ring_wait")) {
        if (G.dev && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_RING_WAIT, value)) return hip_fail();
        G.opt_ring_wait = value != 0;
        return 0;
    }
    if (!strcmp(key, "fir_launch")) {
        if (G.dev && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_FIR_LAUNCH, value)) return hip_fail();
        G.opt_fir_launch = value; G.opt_fir_launch_set = 1;
        return 0;
    }
    if (!strcmp(key, "profile_stride")) {
        if (value < 1) return fail(-1, "profile_stride: every n-th launch, n >= 1");
        if (G.dev && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_PROFILE_STRIDE, value)) return hip_fail();
        G.opt_profile_stride = value;
        return 0;
    }
    if (!strcmp(key, "ready_timeouts")) {                /* 0: the caller has heard of them (and has re-uploaded a state it trusts) */
        if (value) return fail(-1, "ready_timeouts can only be set to 0 (acknowledge)");
        if (G.dev && avdsp_hip_ready_clear(G.dev)) return hip_fail();
        return 0;
    }
    if (!strcmp(key, "group_fanout")) {                  /* 0: a plan's cascade launches (one per section count) one after the other, as through round 4 */
        if (G.dev && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_GROUP_FANOUT, value)) return hip_fail();
        G.opt_group_serial = !value;
        return 0;
    }
    if (!strcmp(key, "cu_split")) {                      /* experiment: the cascades' stream on `value` CUs of its own (applies to the program's device copy as it is) */
        if (G.dev && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_CU_SPLIT, value)) return hip_fail();
        G.opt_cu_split = value;
        return 0;
    }
    if (!strcmp(key, "ready_test")) {                    /* tests only (AVDSP_OPT_READY_TEST) */
        if (G.dev && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_READY_TEST, value)) return hip_fail();
        return 0;
    }
    if (!strcmp(key, "profile")) {
        G.opt_profile = value;
        if (G.dev) avdsp_hip_profile_enable(G.dev, value);
        return 0;
    }
    return fail(-1, "unknown option '%s'", key);
}

/* An option goes to the current program AND becomes the default of programs loaded later (one program per process, the
 * usual case, then behaves as if options were process-wide). */
int dspRuntimeSetOption(const char *key, int value)
{
    int rc = set_option_here(key, value);
    if (!rc && g_cur != &g_template) {
        avdsp_ctx *keep = g_cur;
        g_cur = &g_template;
        rc = set_option_here(key, value);
        g_cur = keep;
    }
    return rc;
}

int dspRuntimeGetOption(const char *key)
{
    if (!strcmp(key, "fir_impl"))    return G.opt_fir_impl;
    if (!strcmp(key, "biquad_impl")) return G.opt_biquad_impl;
    if (!strcmp(key, "device"))      return G.opt_device;
    if (!strcmp(key, "profile"))     return G.opt_profile;
    if (!strcmp(key, "generic"))     return G.opt_generic;
    if (!strcmp(key, "interp_impl")) return G.opt_interp_impl;
    if (!strcmp(key, "levels"))      return G.last_levels;
    if (!strcmp(key, "cores"))       return G.last_cores;
    if (!strcmp(key, "pieces"))      return G.last_pieces;
    if (!strcmp(key, "strand_split")) return G.opt_strand_split;
    if (!strcmp(key, "strand_lanes")) return G.opt_strand_lanes;
    if (!strcmp(key, "strands"))     return G.last_strands;
    if (!strcmp(key, "overlap"))     return G.opt_overlap;
    if (!strcmp(key, "ready_words")) return G.opt_ready_words;
    if (!strcmp(key, "lane_hw"))     return G.opt_lane_hw;
    if (!strcmp(key, "fir_split"))   return G.opt_fir_split;
    if (!strcmp(key, "fir_launch"))  return G.opt_fir_launch_set ? G.opt_fir_launch : -1;
    if (!strcmp(key, "fir_lean"))    return G.opt_fir_lean_set ? G.opt_fir_lean : -1;
    if (!strcmp(key, "ring_wait"))   return G.opt_ring_wait;
    if (!strncmp(key, "timing_pairs_", 13) && key[13] >= '0' && key[13] <= '7' && !key[14])      /* of the latest dspRuntimeKernelTime(kind) */
        return G.dev ? avdsp_hip_profile_last_pairs(G.dev, key[13] - '0') : 0;
    if (!strcmp(key, "ready_timeouts")) { device_current(); return G.dev ? avdsp_hip_ready_timeouts(G.dev) : 0; }
    if (!strcmp(key, "cu_split"))    return G.opt_cu_split;
    if (!strcmp(key, "group_fanout")) return !G.opt_group_serial;
    if (!strcmp(key, "side_by_side")) { device_current(); return G.dev ? avdsp_hip_prog_get_option(G.dev, AVDSP_OPT_SIDE_BY_SIDE) : -1; }
    if (!strcmp(key, "streams_remade")) { device_current(); return G.dev ? avdsp_hip_prog_get_option(G.dev, AVDSP_OPT_STREAMS_REMADE) : 0; }
    if (!strcmp(key, "ready_mode"))  { device_current(); return G.dev ? avdsp_hip_prog_get_option(G.dev, AVDSP_OPT_READY_MODE) : 0; }
    if (!strcmp(key, "fir_rows"))    return G.opt_fir_rows;
    if (!strcmp(key, "host_split"))  return G.opt_host_split;
    if (!strcmp(key, "host_pin"))    return G.opt_host_pin;
    if (!strcmp(key, "shard_rank"))  return G.shard_rank;
    if (!strcmp(key, "shard_world")) return G.shard_world;
    return -1;
}

/* ---- channel sharding ----
 * The chains of a core are independent (lower_core proves it per core), so N processes -- one per GPU -- may each
 * run a contiguous range of them with no exchange at all.  Contiguous and balanced: the first (total % world)
 * ranks take one chain more, i.e. avdsp_amd/sharding.py shard_range().                                        */
static void shard_range(int total, int world, int rank, int *lo, int *hi)
{
    const int q = total / world, r = total % world;
    *lo = rank * q + (rank < r ? rank : r);
    *hi = *lo + q + (rank < r ? 1 : 0);
}

int dspRuntimeSetShard(int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) return fail(-1, "shard %d of %d: need 0 <= rank < world", rank, world);
    g_template.shard_rank = rank; g_template.shard_world = world;       /* programs loaded later start with it (as options do) */
    if (rank == G.shard_rank && world == G.shard_world) return 0;
    G.shard_rank = rank; G.shard_world = world;
    return replan();                     /* the FIR histories go back into the mirror; cores are lowered again */
}

/* ---- dsp_runtime.c:42-59 ---- */
opcode_t *dspFindCore(opcode_t *codePtr, const int numCore)
{
    (void)ctx_of(codePtr);
    if (codePtr->op.opcode != DSP_HEADER) return 0;
    opcode_t *p = codePtr;
    int seen = 0;
    /* the reference follows the skips wherever they lead; inside the loaded program the walk is kept inside it
     * (a damaged skip ends the search instead of reading past the caller's buffer) */
    const opcode_t *end = (dspHeaderPtr && codePtr == G.code) ? G.code + dspHeaderPtr->totalLength : 0;
    for (;;) {
        if (end && p >= end) return 0;
        unsigned skip = p->op.skip;
        if (skip == 0) return seen == 0 ? codePtr : 0;
        if (p->op.opcode == DSP_CORE && ++seen == numCore) return p;
        p += skip;
    }
}

/* ---- dsp_runtime.c:62-77 ---- */
opcode_t *dspFindCoreBegin(opcode_t *p)
{
    if (p) (void)ctx_of(p);
    const opcode_t *end = (dspHeaderPtr && G.code && p >= G.code && p < G.code + dspHeaderPtr->totalLength)
                              ? G.code + dspHeaderPtr->totalLength : 0;        /* as in dspFindCore */
    if (p && p->op.opcode == DSP_CORE)
        for (;;) {
            unsigned skip = p->op.skip;
            int op = p->op.opcode;
            if (skip == 0 || (end && p + skip >= end)) return p;
            if (op == DSP_CORE || op == DSP_NOP || op == DSP_PARAM || op == DSP_PARAM_NUM) p += skip;
            else break;
        }
    return p;
}

/* ---- dsp_runtime.c:116-145 ---- */
int dspRuntimeReset(const int fs, int random, int defaultDither)
{
    if (!dspHeaderPtr) return fail(-1, "dspRuntimeReset before dspRuntimeInit");
    device_current();
    int idx = dspConvertFrequencyToIndex(fs);
    if (idx >= FMAXpos) return fail(-1, "sampling frequency %d not supported", fs);
    int mn = dspHeaderPtr->freqMin, mx = dspHeaderPtr->freqMax;
    if (idx < mn || idx > mx) return fail(-2, "sampling frequency %d outside the program's range", fs);
    G.freq_index = idx - mn;
    G.num_freq = mx - mn + 1;
    g_rate_static = G.num_freq;
    dspBiquadFreqSkip = 2 + 6 * G.num_freq;
    G.biquad_offset = 5 + 6 * G.freq_index;
    G.have_rate = 1;
    G.fs_index = idx;
    G.random = random;                   /* seeds the device copy of the TPDF generator at first use */
    G.dither = defaultDither;
    /* dsp_tpdf.h:55-59: mask = -1 << (32 - dither) */
    G.store_mask = (int)(0xFFFFFFFFu << ((32 - defaultDither) & 31));
    /* the reference zeroes the data area only (:141): words that DSP_STORE_MEM wrote into the program's
     * parameter sections survive a reset, so fetch them before the device copy is dropped */
    if (G.dev && avdsp_hip_ready_clear(G.dev)) return hip_fail();      /* (a reset acknowledges ready-word time-outs: the state they spoiled is zeroed below) */
    if (G.dev && G.dev_state_valid) {
        const int first = (int)(sizeof(dspHeader_t) / sizeof(int));
        if (avdsp_hip_download_words(G.dev, (int32_t *)G.code, first, dspHeaderPtr->totalLength - first))
            return hip_fail();
    }
    int *data = (int *)dspHeaderPtr + dspHeaderPtr->totalLength;
    memset(data, 0, (size_t)dspHeaderPtr->dataSize * sizeof(int));
    /* plans embed the rate-dependent coefficient addresses and the store mask: rebuild lazily */
    drop_device();
    return 0;
}

/* ---- dsp_runtime.c:150-195 ---- */
int dspRuntimeInit(opcode_t *codePtr, int maxSize, const int fs, int random, int defaultDither)
{
    /* the program's context: the one this buffer already has (loaded again: it starts clean), or a new one with the
     * options in force now */
    avdsp_ctx *c = 0, *prev = g_cur;
    for (int i = 0; i < g_nctx; i++) if (g_ctx[i]->code == codePtr) c = g_ctx[i];
    /* a program that overlaps this buffer can only be one whose buffer the caller has freed since: forget it */
    if (codePtr->op.opcode == DSP_HEADER) {
        const dspHeader_t *h = (const dspHeader_t *)codePtr;
        const opcode_t *lo = codePtr, *hi = codePtr + (long long)h->totalLength + (h->dataSize > 0 ? h->dataSize : 0);
        for (int i = 0; i < g_nctx; )
            if (g_ctx[i] != c && g_ctx[i]->code < hi && lo < g_ctx[i]->code + (g_ctx[i]->total_words > 0 ? g_ctx[i]->total_words : 1)) ctx_destroy(g_ctx[i]);
            else i++;
    }
    int fresh = 0;
    if (!c) {
        if (g_nctx == MAX_PROGRAMS) return fail(-9, "%d programs are loaded already: dspRuntimeReleaseProgram() the ones no longer used", MAX_PROGRAMS);
        c = (avdsp_ctx *)calloc(1, sizeof *c);
        if (!c) return fail(-9, "out of memory");
        const avdsp_ctx *o = &g_template;
        c->opt_fir_impl = o->opt_fir_impl; c->opt_biquad_impl = o->opt_biquad_impl; c->opt_device = o->opt_device; c->opt_profile = o->opt_profile;
        c->opt_generic = o->opt_generic; c->opt_interp_impl = o->opt_interp_impl; c->opt_strand_split = o->opt_strand_split; c->opt_strand_lanes = o->opt_strand_lanes;
        c->opt_profile_stride = o->opt_profile_stride; c->opt_overlap = o->opt_overlap; c->opt_fir_rows = o->opt_fir_rows; c->opt_host_split = o->opt_host_split; c->opt_host_pin = o->opt_host_pin; c->opt_ready_words = o->opt_ready_words; c->opt_lane_hw = o->opt_lane_hw; c->opt_fir_split = o->opt_fir_split; c->opt_fir_launch = o->opt_fir_launch; c->opt_fir_launch_set = o->opt_fir_launch_set; c->opt_fir_lean = o->opt_fir_lean; c->opt_fir_lean_set = o->opt_fir_lean_set; c->opt_ring_wait = o->opt_ring_wait; c->opt_cu_split = o->opt_cu_split; c->opt_group_serial = o->opt_group_serial;
        c->shard_rank = o->shard_rank; c->shard_world = o->shard_world;
        c->mantissa = DSP_MANT; c->device_ordinal = -1;
        c->code = codePtr;
        g_ctx[g_nctx++] = c;
        fresh = 1;
    }
    ctx_make_current(c);
    drop_device();
    G.have_rate = 0;
    dspHeaderPtr = (dspHeader_t *)codePtr;
    G.code = codePtr;
    G.total_words = 0;
    /* a buffer that holds no loadable program leaves no context behind (a host probing files would otherwise fill the table) */
#define INIT_FAIL(...) do { const int r_ = fail(__VA_ARGS__); if (fresh) { ctx_destroy(c); for (int i_ = 0; i_ < g_nctx; i_++) if (g_ctx[i_] == prev) ctx_make_current(prev); } return r_; } while (0)
    if (codePtr->op.opcode != DSP_HEADER) INIT_FAIL(-1, "no dsp header in this program");
    int length = dspHeaderPtr->totalLength, size = dspHeaderPtr->dataSize;
    if (size + length > maxSize)
        INIT_FAIL(-6, "program+data = %d words exceeds the buffer (%d)", size + length, maxSize);
    unsigned sum; int cores;
    avdspChecksumWalk(codePtr, (unsigned)length, &sum, &cores);
    if (cores < 1) INIT_FAIL(-3, "no cores defined in the program");
    if (sum != dspHeaderPtr->checkSum) INIT_FAIL(-4, "checksum problem with the program");
    if (dspHeaderPtr->maxOpcode >= DSP_MAX_OPCODE)
        INIT_FAIL(-5, "program uses opcodes newer than this runtime");
#undef INIT_FAIL
    dspMantissa = DSP_MANT;
    /* dsp_runtime.c:181-190 converts the parameters to the runtime's encoding right here.  This library serves every
     * DSP_FORMAT, so which encoding is wanted is only known when the first entry point is called: the conversion runs
     * then (ensure_encoding), with the rate count the reference would have used at THIS point -- the one the previous
     * dspRuntimeReset of the process left behind, 0 the first time (dspNumSamplingFreq is a static that Init does not
     * touch, :106,131), which is why a freshly loaded program's biquad banks stay unconverted there and here. */
    G.numfreq_at_init = g_rate_static;
    G.total_words = length + size;
    if (fs) { int r = dspRuntimeReset(fs, random, defaultDither); if (r) return r; }
    return length;
}

/* ------------------------------------------------------------------------------------------
 * dspChangeFormat (dsp_runtime.c:198-299): rewrite the encoded parameters in place, Q(old) -> Q(new), Q -> float or
 * float -> Q, walking the opcode stream from the header to the end of the program.  Which words are rewritten follows
 * the reference opcode by opcode, its blind spots included: nothing for DSP_FIR (":267 TODO"), only the first word of a
 * DATA_TABLE, and per-rate tables only as far as the rate count it happens to know (see dspRuntimeInit).  The reference
 * follows every offset on trust; here a target outside the program words is refused (-8) before anything is written.
 * ---------------------------------------------------------------------------------------- */
static void change_datum(opcode_t *w, int oldf, int newf)                    /* dspChangeThisData, :198-209 */
{
    if (oldf) {
        if (newf) {
            int d = newf - oldf;
            if (d > 0) w->u32 <<= d;
            if (d < 0) w->i32 >>= -d;
        } else w->f32 = (float)w->i32 / (float)(1 << oldf);
    } else if (newf) w->i32 = (int)dspQNM(w->f32, 32 - newf, newf);          /* the call of :208 with the argument it lacks */
}

static int change_format(int newf, int numfreq)
{
    const int total = dspHeaderPtr->totalLength, oldf = dspHeaderPtr->format;
    /* two passes over the same walk: check every target, then write */
    for (int pass = 0; pass < 2; pass++) {
        opcode_t *p = G.code;
        for (;;) {
            const int at = (int)(p - G.code);
            if (at < 0 || at >= total) return fail(-8, "opcode stream runs past the program (word %d)", at);
            const unsigned skip = p->op.skip;
            if (skip == 0) break;
            if ((long long)at + skip > total) return fail(-8, "word %d: opcode longer than the program", at);
#define CF_AT(ptr) do { long long w_ = (ptr) - G.code; if (w_ < 12 || w_ >= total) return fail(-8, "word %d: parameter at word %lld outside the program", at, w_); \
                        if (pass) change_datum((ptr), oldf, newf); } while (0)
#define CF_NEED(n) do { if ((unsigned)(1 + (n)) > skip) return fail(-8, "word %d: opcode payload shorter than %d words", at, (n)); } while (0)
            opcode_t *a = p + 1;
            switch (p->op.opcode) {
            case DSP_DIRAC: case DSP_SQUAREWAVE:
                CF_NEED(2); a++;                       /* immediate behind a data pointer */
                CF_AT(a); break;
            case DSP_MUL_VALUE: case DSP_DIV_VALUE: case DSP_DATA_TABLE: case DSP_CLIP:
                CF_NEED(1); CF_AT(a); break;
            case DSP_LOAD_GAIN:
                CF_NEED(2); a++;                       /* IO number, then the gain's offset */
                CF_AT(p + a->i32); break;
            case DSP_GAIN: case DSP_SAT0DB_GAIN: case DSP_SAT0DB_TPDF_GAIN:
                CF_NEED(1); CF_AT(p + a->i32); break;
            case DSP_LOAD_MUX: {
                CF_NEED(1);
                opcode_t *t = p + a->i32;
                if (t - G.code < 12 || t - G.code >= total) return fail(-8, "word %d: mux table outside the program", at);
                const int n = (short)t->i32;
                t++;
                for (int i = 0; i < n; i++) { t++; CF_AT(t); t++; }
                break; }
            case DSP_BIQUADS: {
                CF_NEED(2); a++;
                opcode_t *t = p + a->i32;
                if (t - G.code < 12 || t - G.code >= total) return fail(-8, "word %d: biquad bank outside the program", at);
                const int ns = (short)t->i32;
                t += 3;
                for (int i = 0; i < ns; i++) {
                    t += 2;
                    for (int j = 0; j < numfreq; j++) { for (int k = 0; k < 5; k++) { CF_AT(t); t++; } t++; }
                }
                break; }
            case DSP_DITHER_NS2: {
                CF_NEED(2); a++;
                opcode_t *t = p + a->i32;
                for (int i = 0; i < 3 * numfreq; i++) { CF_AT(t); t++; }
                break; }
            case DSP_DCBLOCK:
                a++;
                for (int i = 0; i < numfreq; i++) { CF_NEED(2 + i); CF_AT(a); a++; }
                break;
            case DSP_SINE:
                CF_NEED(2); a++;
                CF_AT(a); a++;
                for (int i = 0; i < numfreq; i++) { CF_NEED(3 + i); CF_AT(a); a++; }
                break;
            default: break;
            }
#undef CF_AT
#undef CF_NEED
            p += skip;
        }
    }
    dspHeaderPtr->format = (unsigned short)newf;
    return 0;
}

/* the program's parameters in the encoding the entry point DSP_FORMAT `format` computes with */
static int ensure_encoding(int format)
{
    const int want = format == DSP_FORMAT_INT64 ? DSP_MANT : 0;
    if (dspHeaderPtr->format == want) return 0;
    /* the program words change under the device's feet: bring its state home first, rebuild it afterwards */
    if (G.dev && G.dev_state_valid) {
        const int first = (int)(sizeof(dspHeader_t) / sizeof(int));
        if (avdsp_hip_download_words(G.dev, (int32_t *)G.code, first, G.total_words - first)) return hip_fail();
    }
    drop_device();
    return change_format(want, G.numfreq_at_init);
}

/* ------------------------------------------------------------------------------------------
 * lowering: opcode stream of one core -> chains
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    avdsp_chain *chains; int nchains, cap_chains;
    int32_t *coef_word, *state_word; int nsec, cap_sec;
} lowered;

static void lowered_free(lowered *L) { free(L->chains); free(L->coef_word); free(L->state_word); memset(L, 0, sizeof *L); }

static int push_chain(lowered *L, const avdsp_chain *c)
{
    if (L->nchains == L->cap_chains) {
        int n = L->cap_chains ? 2 * L->cap_chains : 64;
        avdsp_chain *p = (avdsp_chain *)realloc(L->chains, (size_t)n * sizeof *p);
        if (!p) return -1;
        L->chains = p; L->cap_chains = n;
    }
    L->chains[L->nchains++] = *c;
    return 0;
}

static int push_section(lowered *L, int coef, int state)
{
    if (L->nsec == L->cap_sec) {
        int n = L->cap_sec ? 2 * L->cap_sec : 1024;
        int32_t *a = (int32_t *)realloc(L->coef_word, (size_t)n * 4);
        int32_t *b = (int32_t *)realloc(L->state_word, (size_t)n * 4);
        if (!a || !b) { if (a) L->coef_word = a; if (b) L->state_word = b; return -1; }
        L->coef_word = a; L->state_word = b; L->cap_sec = n;
    }
    L->coef_word[L->nsec] = coef;
    L->state_word[L->nsec] = state;
    L->nsec++;
    return 0;
}

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }

/* Chains of one core run concurrently on the device, so the sequential reading of the opcode
 * stream must not create a dependency between them: no chain may load an IO that another chain
 * of the same core stores, and no IO may be stored twice.                                       */
static int check_independent(const lowered *L)
{
    int nout = 0;
    for (int i = 0; i < L->nchains; i++) nout += L->chains[i].n_out;
    int *outs = (int *)malloc((size_t)(nout ? nout : 1) * sizeof(int));
    if (!outs) return fail(-9, "out of memory");
    int k = 0;
    for (int i = 0; i < L->nchains; i++)
        for (int j = 0; j < L->chains[i].n_out; j++) outs[k++] = L->chains[i].out_io[j];
    qsort(outs, (size_t)nout, sizeof(int), cmp_int);
    int rc = 0;
    for (int i = 1; i < nout && !rc; i++)
        if (outs[i] == outs[i - 1]) rc = fail(-8, "IO %d is stored by more than one chain of the core", outs[i]);
    for (int i = 0; i < L->nchains && !rc; i++)
        if (bsearch(&L->chains[i].in_io, outs, (size_t)nout, sizeof(int), cmp_int))
            rc = fail(-8, "IO %d is both loaded and stored inside one core (cross-chain dependency)", L->chains[i].in_io);
    free(outs);
    return rc;
}

#define GENERIC_IO_LIMIT 65536            /* IO numbers a program may use (the reference's hosts use well under 100) */

static int lower_core(int format, opcode_t *core, lowered *L)
{
    const int float_alu = (format != DSP_FORMAT_INT64);
    const int prog_words = dspHeaderPtr->totalLength;
    opcode_t *p = dspFindCoreBegin(core);
    avdsp_chain cur;
    int open = 0;
    memset(&cur, 0, sizeof cur);
    memset(L, 0, sizeof *L);

    if (ensure_encoding(format)) return g_err_code;

    for (;;) {
        const int op = p->op.opcode;
        const unsigned skip = p->op.skip;
        const int *a = (const int *)p + 1;
        const int at = (int)(p - G.code);
        if (at < 0 || at >= prog_words) return fail(-8, "opcode stream runs past the program (word %d)", at);
        if (skip == 0 || op == DSP_CORE) break;              /* dsp_runtime.c:321-331 */
        if ((long long)at + skip > prog_words) return fail(-8, "word %d: opcode longer than the program", at);
        /* the encoder is trusted by the reference; here nothing is dereferenced before it is known to lie inside
         * the program (the device library checks the word indices of the finished plan once more) */
#define LC_NEED(n) do { if ((unsigned)(1 + (n)) > skip) return fail(-8, "word %d: opcode payload shorter than %d words", at, (n)); } while (0)
#define LC_PROG(off, n) do { long long lo_ = (long long)at + (off); if (lo_ < 0 || lo_ + (n) > prog_words) \
            return fail(-8, "word %d: program offset %d (+%d) outside the program", at, (int)(off), (int)(n)); } while (0)
        switch (op) {
        case DSP_NOP: case DSP_PARAM: case DSP_PARAM_NUM: case DSP_SERIAL:
            break;
        case DSP_LOAD:                                        /* :565-583 */
        case DSP_LOAD_GAIN:                                   /* :586-607 */
            LC_NEED(op == DSP_LOAD_GAIN ? 2 : 1);
            if (op == DSP_LOAD_GAIN) LC_PROG(a[1], 1);
            if (a[0] < 0 || a[0] >= GENERIC_IO_LIMIT) return fail(-8, "word %d: IO number %d outside [0,%d)", at, a[0], GENERIC_IO_LIMIT);
            if (open) {
                if (cur.n_out == 0)
                    return fail(-8, "word %d: value replaced before any STORE (X/Y tricks are not lowered)", at);
                if (push_chain(L, &cur)) return fail(-9, "out of memory");
            }
            memset(&cur, 0, sizeof cur);
            open = 1;
            cur.in_io = a[0];
            cur.sec_base = L->nsec;
            if (op == DSP_LOAD_GAIN) {
                cur.load_mode = AVDSP_LOAD_GAIN;
                cur.gain_bits = p[a[1]].u32;
            }
            break;
        case DSP_BIQUADS: {                                   /* :827-849 */
            if (!open || cur.n_out || cur.sat || cur.fir_taps)
                return fail(-8, "word %d: BIQUADS outside the supported LOAD->BIQUADS->FIR->SAT0DB->STORE order", at);
            LC_NEED(2);
            LC_PROG(a[1], 2);
            const opcode_t *bank = p + a[1];
            int num = (short)bank[0].i32;                     /* callee takes `short num` */
            if (num < 1) return fail(-8, "word %d: biquad bank with %d sections", at, num);
            LC_PROG(a[1], G.biquad_offset + (num - 1) * dspBiquadFreqSkip + 5);
            if (a[0] < 0 || (long long)a[0] + 6ll * num > dspHeaderPtr->dataSize)
                return fail(-8, "word %d: data offset %d (+%d) outside the state area (%d words)", at, a[0], 6 * num, dspHeaderPtr->dataSize);
            if (bank[1].i32 != 0)                             /* 0 = bypass, :837 */
                for (int s = 0; s < num; s++)
                    if (push_section(L, (int)(bank - G.code) + G.biquad_offset + s * dspBiquadFreqSkip,
                                     prog_words + a[0] + 6 * s))
                        return fail(-9, "out of memory");
            cur.nsec = L->nsec - cur.sec_base;
            break; }
        case DSP_FIR: {                                       /* :928-969 */
            if (!open || cur.n_out || cur.sat)
                return fail(-8, "word %d: FIR outside the supported chain order", at);
            if (!float_alu)
                return fail(-8, "word %d: DSP_FIR in int64 mode is undefined behaviour in the reference "
                                "(dsp_firSTD.h:8-35) and is not provided", at);
            LC_NEED(G.num_freq + 1);
            int off = a[G.freq_index];
            if (off) {
                LC_PROG(off, 1);
                const opcode_t *imp = p + off;
                int length = imp[0].i32;
                if (length >> 16)
                    return fail(-8, "word %d: FIR pure-delay variant is not lowered yet", at);
                if (length > 0) {
                    LC_PROG(off, 1 + length);
                    if (a[G.num_freq] < 0 || (long long)a[G.num_freq] + length > dspHeaderPtr->dataSize)
                        return fail(-8, "word %d: data offset %d (+%d) outside the state area (%d words)", at, a[G.num_freq], length, dspHeaderPtr->dataSize);
                    if (cur.fir_taps) return fail(-8, "word %d: more than one FIR per chain", at);
                    cur.fir_taps = length;
                    cur.fir_coef_word = (int)(imp - G.code) + 1;
                    cur.fir_state_word = prog_words + a[G.num_freq];
                }
            }
            break; }
        case DSP_SAT0DB:                                      /* :464-475 */
            if (!open || cur.n_out) return fail(-8, "word %d: SAT0DB outside the supported chain order", at);
            cur.sat = 1;
            break;
        case DSP_STORE:                                       /* :610-633 */
            LC_NEED(1);
            if (a[0] < 0 || a[0] >= GENERIC_IO_LIMIT) return fail(-8, "word %d: IO number %d outside [0,%d)", at, a[0], GENERIC_IO_LIMIT);
            if (!open) return fail(-8, "word %d: STORE without a LOAD", at);
            if (cur.n_out == AVDSP_MAX_STORES) return fail(-8, "word %d: more than %d STOREs in a chain", at, AVDSP_MAX_STORES);
            cur.out_io[cur.n_out++] = a[0];
            break;
        default:
            return fail(-8, "word %d: opcode %d (%s) is not lowered to the HIP path", at, op,
                        (op >= 0 && op < DSP_MAX_OPCODE) ? dspOpcodeText[op] + (dspOpcodeText[op][0] == '\n') : "?");
        }
        p += skip;
    }
    if (open) {
        if (cur.n_out == 0) return fail(-8, "core ends with a chain that is never stored");
        if (push_chain(L, &cur)) return fail(-9, "out of memory");
    }
#undef LC_NEED
#undef LC_PROG
    if (L->nchains == 0) return fail(-8, "core contains no LOAD..STORE chain");
    return check_independent(L);
}


/* ------------------------------------------------------------------------------------------
 * general path: bounds-check every offset the interpreter will follow (dsp_runtime.c trusts the
 * encoder; a device kernel must not), collect the IO span, refuse what has no defined result
 * ---------------------------------------------------------------------------------------- */
/* Frame-parallel eligibility (avdsp_interp.inc, interp_wave): the device may run 64 frames of a block side
 * by side, opcode by opcode, only if nothing but opcode-private state passes from one frame to the next.
 * While the opcodes are walked in program order this records what a frame reads before it has written it:
 *  - frame slots (samples[]) that are also stored later in the frame: they must then come from the caller's
 *    windows -- decided per block call, the slots are handed over as a bitmap;
 *  - mirror words used as per-frame variables (STORE_MEM / LOAD_MEM, the result words of LOAD_MUX, TPDF,
 *    TPDF_CALC read by LOAD_MEM_DATA): written-then-read words get a per-lane copy; a word read first and
 *    written later in the frame is last frame's value = not eligible;
 *  - the TPDF generator: DSP_TPDF_CALC must run before anything that uses its value or the global mask;
 *  - private state areas of two opcodes must not overlap (the encoder never shares them).               */
#define WAVE_IO_LIMIT 256
typedef struct {
    int ok, complete;
    unsigned char written[WAVE_IO_LIMIT], early[WAVE_IO_LIMIT];
    int nwr, wr_word[64];
    int nearly, early_word[64];
    int nranges; struct { long long lo, hi; } range[512];
    int nparams; int param[512][2];      /* program words read as parameters (gains, banks, tables, taps ...), [lo, hi) */
    int tpdf_calc_seen, tpdf_user_seen;
} wscan;

typedef struct {
    int format, aw, prog_words, data_words, at;
    unsigned skip;
    avdsp_generic_desc *d;
    wscan w;
} gscan;

static void ws_read_io(gscan *s, int io)
{
    if (io < 0 || io >= WAVE_IO_LIMIT) { s->w.ok = s->w.complete = 0; return; }
    if (!s->w.written[io]) s->w.early[io] = 1;
}
static void ws_write_io(gscan *s, int io)
{
    if (io < 0 || io >= WAVE_IO_LIMIT) { s->w.ok = s->w.complete = 0; return; }
    s->w.written[io] = 1;
}
static void ws_mem_write(gscan *s, int word)
{
    for (int k = 0; k < s->w.nwr; k++) if (s->w.wr_word[k] == word) return;
    if (s->w.nwr == 64) { s->w.ok = s->w.complete = 0; return; }
    s->w.wr_word[s->w.nwr++] = word;
}
static void ws_mem_read(gscan *s, int word, int needs_writer)
{
    for (int k = 0; k < s->w.nwr; k++)
        if (s->w.wr_word[k] == word) {
            avdsp_generic_desc *d = s->d;
            for (int j = 0; j < d->nvm; j++) if (d->vm_word[j] == word) return;
            if (d->nvm == 16) { s->w.ok = 0; return; }
            d->vm_word[d->nvm++] = word;
            return;
        }
    if (needs_writer) s->w.ok = 0;                   /* LOAD_MEM_DATA of a word nothing in this frame has written yet */
    for (int k = 0; k < s->w.nearly; k++) if (s->w.early_word[k] == word) return;
    if (s->w.nearly == 64) { s->w.ok = s->w.complete = 0; return; }
    s->w.early_word[s->w.nearly++] = word;
}
static void ws_state(gscan *s, int off, long long n)
{
    if (n <= 0) return;
    if (s->w.nranges == 512) { s->w.ok = s->w.complete = 0; return; }
    s->w.range[s->w.nranges].lo = off; s->w.range[s->w.nranges].hi = off + n; s->w.nranges++;
}
static void ws_tpdf_user(gscan *s) { s->w.tpdf_user_seen = 1; }
static void ws_finish(gscan *s)
{
    wscan *w = &s->w;
    /* a slot read first and stored later in the frame carries last frame's value, unless the caller's rows
     * overwrite it every frame (per call); a slot this core only reads is constant during its block */
    for (int io = 0; io < WAVE_IO_LIMIT; io++)
        if (w->early[io] && w->written[io]) s->d->carried_io[io >> 5] |= 1u << (io & 31);
    for (int i = 0; i < w->nearly && w->ok; i++)
        for (int k = 0; k < w->nwr; k++) if (w->wr_word[k] == w->early_word[i]) { w->ok = 0; break; }
    /* a word this core stores with STORE_MEM and an opcode of it reads as a parameter (a gain computed on the fly):
     * that opcode must see this frame's value, or last frame's, depending on the order -- frame by frame only */
    for (int i = 0; i < w->nwr && w->ok; i++)
        for (int k = 0; k < w->nparams; k++)
            if (w->wr_word[i] < w->param[k][1] && w->wr_word[i] + s->aw > w->param[k][0]) { w->ok = 0; break; }
    /* values are aw words wide: two different keys closer than that would alias */
    for (int i = 0; i < w->nwr && w->ok; i++) {
        for (int k = 0; k < w->nwr; k++) { int dlt = w->wr_word[i] - w->wr_word[k]; if (dlt && dlt > -s->aw && dlt < s->aw) w->ok = 0; }
        for (int k = 0; k < w->nearly; k++) { int dlt = w->wr_word[i] - w->early_word[k]; if (dlt && dlt > -s->aw && dlt < s->aw) w->ok = 0; }
    }
    for (int i = 0; i < w->nranges && w->ok; i++)
        for (int k = i + 1; k < w->nranges; k++)
            if (w->range[i].lo < w->range[k].hi && w->range[k].lo < w->range[i].hi) { w->ok = 0; break; }
    /* a result word (data area) inside some opcode's private state would be the same kind of sharing */
    for (int i = 0; i < w->nwr && w->ok; i++) {
        long long off = (long long)w->wr_word[i] - s->prog_words;
        if (off < 0) continue;
        for (int k = 0; k < w->nranges; k++) if (off < w->range[k].hi && off + s->aw > w->range[k].lo) { w->ok = 0; break; }
    }
}

static int cmp_range(const void *a, const void *b)
{
    const int *x = (const int *)a, *y = (const int *)b;
    return (x[0] > y[0]) - (x[0] < y[0]);
}

/* what the core touches, for the device (what a launch writes back) and for dspRuntimeBlockAll (which cores may
 * run side by side); state areas are merged: the encoder allocates them back to back */
static void ws_export(const gscan *s, core_deps *dp)
{
    const wscan *w = &s->w;
    memset(dp, 0, sizeof *dp);
    dp->complete = w->complete;
    for (int io = 0; io < WAVE_IO_LIMIT; io++) {
        if (w->early[io]) dp->early_io[io >> 5] |= 1u << (io & 31);
        if (w->written[io]) dp->written_io[io >> 5] |= 1u << (io & 31);
    }
    dp->nwr = w->nwr; memcpy(dp->wr_word, w->wr_word, sizeof dp->wr_word);
    dp->nrd = w->nearly; memcpy(dp->rd_word, w->early_word, sizeof dp->rd_word);
    dp->tpdf_calc = w->tpdf_calc_seen; dp->tpdf_user = w->tpdf_user_seen;
    static int tmp[512][2];
    for (int i = 0; i < w->nranges; i++) { tmp[i][0] = s->prog_words + (int)w->range[i].lo; tmp[i][1] = s->prog_words + (int)w->range[i].hi; }
    qsort(tmp, (size_t)w->nranges, sizeof tmp[0], cmp_range);
    int n = 0;
    for (int i = 0; i < w->nranges; i++) {
        if (n && tmp[i][0] <= dp->range[n - 1][1]) { if (tmp[i][1] > dp->range[n - 1][1]) dp->range[n - 1][1] = tmp[i][1]; continue; }
        if (n == DEPS_MAX_RANGES) { dp->complete = 0; break; }
        dp->range[n][0] = tmp[i][0]; dp->range[n][1] = tmp[i][1]; n++;
    }
    dp->nranges = n;
    static int ptmp[512][2];
    memcpy(ptmp, w->param, sizeof(int) * 2 * (size_t)w->nparams);
    qsort(ptmp, (size_t)w->nparams, sizeof ptmp[0], cmp_range);
    n = 0;
    for (int i = 0; i < w->nparams; i++) {
        if (n && ptmp[i][0] <= dp->param[n - 1][1]) { if (ptmp[i][1] > dp->param[n - 1][1]) dp->param[n - 1][1] = ptmp[i][1]; continue; }
        if (n == DEPS_MAX_RANGES) { dp->complete = 0; break; }
        dp->param[n][0] = ptmp[i][0]; dp->param[n][1] = ptmp[i][1]; n++;
    }
    dp->nparams = n;
}

static int gs_payload(const gscan *s, int need)
{
    if ((unsigned)(1 + need) > s->skip) return fail(-8, "word %d: opcode payload shorter than %d words", s->at, need);
    return 0;
}
static int gs_prog(const gscan *s, int off, int n)          /* words [at+off, at+off+n) of the program */
{
    long long lo = (long long)s->at + off;
    if (n < 0 || lo < 0 || lo + n > s->prog_words) return fail(-8, "word %d: program offset %d (+%d) outside the program", s->at, off, n);
    return 0;
}
static int gs_param(gscan *s, int off, int n)               /* gs_prog for words an opcode reads as parameters */
{
    if (gs_prog(s, off, n)) return g_err_code;
    wscan *w = &s->w;
    if (n > 0) {
        if (w->nparams == 512) { w->ok = w->complete = 0; return 0; }
        w->param[w->nparams][0] = s->at + off; w->param[w->nparams][1] = s->at + off + n; w->nparams++;
    }
    return 0;
}
static int gs_data(const gscan *s, int off, long long n)    /* words [off, off+n) of the data area */
{
    if (n < 0 || off < 0 || (long long)off + n > s->data_words) return fail(-8, "word %d: data offset %d (+%lld) outside the state area (%d words)", s->at, off, n, s->data_words);
    return 0;
}
static int gs_io(const gscan *s, int io, int is_out)
{
    if (io < 0 || io >= GENERIC_IO_LIMIT) return fail(-8, "word %d: IO number %d outside [0,%d)", s->at, io, GENERIC_IO_LIMIT);
    avdsp_generic_desc *d = s->d;
    if (io + 1 > d->io_span) d->io_span = io + 1;
    if (is_out) { if (io < d->io_out_min) d->io_out_min = io; if (io > d->io_out_max) d->io_out_max = io; }
    else        { if (io < d->io_in_min)  d->io_in_min = io;  if (io > d->io_in_max)  d->io_in_max = io; }
    return 0;
}

/* DSP_STORE_MEM is the one opcode that writes PROGRAM words (dsp_runtime.c:760-766).  The offsets checked here
 * are only worth something if it cannot rewrite what they were checked against, so its target has to be a free
 * word of a PARAM section: not the opcode stream, and not a parameter word some opcode reads as a count, an IO
 * number or a length (biquad bank header, LOAD_MUX table, FIR impulse length).  map: 0 = opcode stream,
 * 1 = free parameter word, 2 = structural parameter word; built over the whole program (all cores).      */
static unsigned char *store_mem_map(void)
{
    const int total = dspHeaderPtr->totalLength, nf = G.num_freq;
    unsigned char *map = (unsigned char *)calloc((size_t)total, 1);
    if (!map) return 0;
    const int *w = (const int *)G.code;
    for (int pass = 0; pass < 2; pass++)
        for (int at = 0; at < total;) {
            const int op = (int)((unsigned)w[at] >> 16);
            const int skip = w[at] & 0xFFFF;
            if (skip == 0 || at + skip > total) break;
            const int *a = w + at + 1;
#define SM_MARK(x) do { long long x_ = (x); if (x_ >= 0 && x_ < total && map[x_] == 1) map[x_] = 2; } while (0)
            if (pass == 0) {
                if (op == DSP_PARAM || op == DSP_PARAM_NUM) memset(map + at + 1, 1, (size_t)(skip - 1));
            } else if (op == DSP_BIQUADS && skip >= 3) {
                SM_MARK((long long)at + a[1]); SM_MARK((long long)at + a[1] + 1);
            } else if (op == DSP_LOAD_MUX && skip >= 3) {
                const long long t = (long long)at + a[0];
                if (t >= 0 && t < total) {
                    const int n = (short)w[t];
                    SM_MARK(t);
                    for (int k = 0; k < n; k++) SM_MARK(t + 1 + 2 * k);
                }
            } else if (op == DSP_FIR && skip >= nf + 2) {
                for (int f = 0; f < nf; f++) if (a[f]) SM_MARK((long long)at + a[f]);
            }
#undef SM_MARK
            at += skip;
        }
    return map;
}

static int scan_generic(int format, opcode_t *core, int end_word, avdsp_generic_desc *d, core_deps *deps)
{
    if (format < 2 || format > 6) return fail(-1, "DSP_FORMAT %d is not one of 2..6", format);
    if (ensure_encoding(format)) return g_err_code;
    const int alu_int = (format == DSP_FORMAT_INT64);
    const int nf = G.num_freq, fi = G.freq_index;
    gscan S;
    unsigned char *smap = 0;                     /* store_mem_map(), built at the first STORE_MEM */
    memset(&S, 0, sizeof S);
    S.format = format; S.aw = (format == 3 || format == 5) ? 1 : 2;
    S.prog_words = dspHeaderPtr->totalLength; S.data_words = dspHeaderPtr->dataSize; S.d = d;
    memset(d, 0, sizeof *d);
    S.w.ok = G.opt_interp_impl != 0;
    S.w.complete = 1;
    d->io_in_min = d->io_out_min = GENERIC_IO_LIMIT; d->io_in_max = d->io_out_max = -1;
    opcode_t *p = end_word ? core : dspFindCoreBegin(core);      /* a strand group starts where it starts */
    d->format = format;
    d->core_word = (int)(p - G.code);
    d->end_word = end_word;
    d->prog_words = S.prog_words;
    d->freq_index = fi; d->num_freq = nf;
    d->biquad_freq_skip = dspBiquadFreqSkip; d->biquad_freq_offset = G.biquad_offset;
    d->delay_line_factor = (unsigned)(4294.967296 * (double)dspConvertFrequencyFromIndex(G.fs_index));   /* dsp_runtime.c:81-90 */

    int nreal = 0;                               /* opcodes that do something */
    for (;;) {
        const int op = p->op.opcode;
        const unsigned skip = p->op.skip;
        const int *a = (const int *)p + 1;
        const int at = (int)(p - G.code);
        if (at < 0 || at >= S.prog_words) { free(smap); return fail(-8, "opcode stream runs past the program (word %d)", at); }
        if (skip == 0 || op == DSP_CORE || op == DSP_END_OF_CODE || at == end_word) break;
        if ((long long)at + skip > S.prog_words) { free(smap); return fail(-8, "word %d: opcode longer than the program", at); }
        S.at = at; S.skip = skip;
        int rc = 0;
        if (op != DSP_NOP && op != DSP_PARAM && op != DSP_PARAM_NUM && op != DSP_SERIAL) nreal++;
        switch (op) {
        case DSP_NOP: case DSP_PARAM: case DSP_PARAM_NUM: case DSP_SERIAL:
        case DSP_SWAPXY: case DSP_COPYXY: case DSP_COPYYX: case DSP_CLRXY:
        case DSP_ADDXY: case DSP_ADDYX: case DSP_SUBXY: case DSP_SUBYX: case DSP_NEGX: case DSP_NEGY:
        case DSP_MULXY: case DSP_DIVXY: case DSP_DIVYX: case DSP_AVGXY: case DSP_AVGYX:
        case DSP_SQRTX: case DSP_SAT0DB:
            break;
        case DSP_SAT0DB_TPDF: case DSP_WHITE:
            ws_tpdf_user(&S); break;
        case DSP_SHIFT: case DSP_MUL_VALUE: case DSP_DIV_VALUE: case DSP_MUL_VALUE_INT: case DSP_DIV_VALUE_INT:
        case DSP_AND_VALUE_INT: case DSP_CLIP:
            rc = gs_payload(&S, 1); break;
        case DSP_LOAD:  rc = gs_payload(&S, 1) || gs_io(&S, a[0], 0); if (!rc) ws_read_io(&S, a[0]); break;
        case DSP_STORE: rc = gs_payload(&S, 1) || gs_io(&S, a[0], 1); if (!rc) { ws_write_io(&S, a[0]); ws_tpdf_user(&S); } break;
        case DSP_LOAD_GAIN: rc = gs_payload(&S, 2) || gs_io(&S, a[0], 0) || gs_param(&S, a[1], 1); if (!rc) ws_read_io(&S, a[0]); break;
        case DSP_LOAD_STORE:
            for (unsigned k = 0; k + 2 <= skip - 1 && !rc; k += 2) {
                rc = gs_io(&S, a[k], 0) || gs_io(&S, a[k + 1], 1);
                if (!rc) { ws_read_io(&S, a[k]); ws_write_io(&S, a[k + 1]); }
            }
            break;
        case DSP_SAT0DB_TPDF_GAIN:
            ws_tpdf_user(&S);
            /* fall through */
        case DSP_GAIN: case DSP_SAT0DB_GAIN: case DSP_VALUE: case DSP_VALUE_INT:
            rc = gs_payload(&S, 1) || gs_param(&S, a[0], 1); break;
        case DSP_LOAD_MEM:
            rc = gs_payload(&S, 1) || gs_prog(&S, a[0], S.aw); if (!rc) ws_mem_read(&S, at + a[0], 0); break;
        case DSP_STORE_MEM:
            rc = gs_payload(&S, 1) || gs_prog(&S, a[0], S.aw);
            if (!rc) {
                if (!smap && !(smap = store_mem_map())) { rc = fail(-9, "out of memory"); break; }
                for (int k = 0; k < S.aw && !rc; k++)
                    if (smap[at + a[0] + k] != 1)
                        rc = fail(-8, "word %d: STORE_MEM target (word %d) is not a free word of a PARAM section: it would "
                                      "rewrite %s", at, at + a[0] + k, smap[at + a[0] + k] ? "a count, IO number or length other opcodes rely on" : "the opcode stream");
            }
            if (!rc) ws_mem_write(&S, at + a[0]);
            break;
        case DSP_TPDF_CALC:
            rc = gs_payload(&S, 2) || gs_data(&S, a[1], S.aw);
            if (!rc) {
                if (S.w.tpdf_calc_seen || S.w.tpdf_user_seen) S.w.ok = 0;
                S.w.tpdf_calc_seen = 1;
                ws_mem_write(&S, S.prog_words + a[1]);
                d->dither_arg = a[0]; d->dither_result_word = S.prog_words + a[1];
            }
            break;
        case DSP_TPDF:
            rc = gs_payload(&S, 2) || gs_data(&S, a[1], S.aw);
            if (!rc) { ws_tpdf_user(&S); ws_mem_write(&S, S.prog_words + a[1]); }
            break;
        case DSP_DELAY_1:
            rc = gs_payload(&S, 1) || gs_data(&S, a[0], S.aw); if (!rc) ws_state(&S, a[0], S.aw); break;
        case DSP_LOAD_MEM_DATA:
            rc = gs_payload(&S, 1) || gs_data(&S, a[0], S.aw); if (!rc) ws_mem_read(&S, S.prog_words + a[0], 1); break;
        case DSP_DELAY: case DSP_DELAY_DP:
            rc = gs_payload(&S, 3);
            if (!rc && a[0] < 0) rc = fail(-8, "word %d: negative delay size", at);
            if (!rc) {
                /* :769-790: with a parameter the first word is the line's size in samples; without
                 * (fixed delay) it is microseconds and the length follows from the sample rate */
                long long n = a[2] ? (long long)a[0] : (long long)(((unsigned long long)(unsigned)a[0] * d->delay_line_factor) >> 32);
                rc = gs_data(&S, a[1], 1 + n * (op == DSP_DELAY ? 1 : S.aw));
                if (!rc) ws_state(&S, a[1], 1 + n * (op == DSP_DELAY ? 1 : S.aw));
            }
            if (!rc && a[2]) rc = gs_param(&S, a[2], 1);
            break;
        case DSP_BIQUADS: {
            rc = gs_payload(&S, 2) || gs_param(&S, a[1], 2);
            if (rc) break;
            const opcode_t *bank = p + a[1];
            const int num = (short)bank[0].i32;
            if (num < 1) { rc = fail(-8, "word %d: biquad bank with %d sections", at, num); break; }
            rc = gs_param(&S, a[1], G.biquad_offset + (num - 1) * dspBiquadFreqSkip + 5) || gs_data(&S, a[0], 6ll * num);
            if (!rc) ws_state(&S, a[0], 6ll * num);
            break; }
        case DSP_LOAD_MUX: {
            rc = gs_payload(&S, 2) || gs_param(&S, a[0], 1) || gs_data(&S, a[1], S.aw);
            if (rc) break;
            const opcode_t *t = p + a[0];
            const int n = (short)t[0].i32;
            if (n > 0) rc = gs_param(&S, a[0], 1 + 2 * n);
            for (int k = 0; k < n && !rc; k++) { rc = gs_io(&S, t[1 + 2 * k].i32, 0); if (!rc) ws_read_io(&S, t[1 + 2 * k].i32); }
            if (!rc) ws_mem_write(&S, S.prog_words + a[1]);
            break; }
        case DSP_DATA_TABLE:
            rc = gs_payload(&S, 5);
            if (!rc && (a[2] < 1 || a[1] < 0 || a[1] > a[2])) rc = fail(-8, "word %d: data table size %d / step %d", at, a[2], a[1]);
            if (!rc) rc = gs_data(&S, a[3], 1) || gs_param(&S, a[4], a[2]);
            if (!rc) ws_state(&S, a[3], 1);
            break;
        case DSP_FIR: {
            rc = gs_payload(&S, nf + 1);
            if (rc) break;
            const int off = a[fi];
            if (!off) break;
            rc = gs_param(&S, off, 1);
            if (rc) break;
            const int length = p[off].i32, delay = length >> 16;
            if (delay) { rc = gs_data(&S, a[nf], 1 + (long long)delay); if (!rc) ws_state(&S, a[nf], 1 + (long long)delay); }
            else if (length > 0) {
                if (alu_int)
                    rc = fail(-8, "word %d: DSP_FIR in int64 mode is undefined behaviour in the reference "
                                  "(dsp_firSTD.h:8-35) and is not provided", at);
                else rc = gs_param(&S, off, 1 + length) || gs_data(&S, a[nf], length);
                if (!rc) { ws_state(&S, a[nf], length); if (length + 64 > d->seq_words) d->seq_words = length + 64; }
            }
            break; }
        case DSP_RMS:
            rc = gs_payload(&S, 2 + 2 * nf);
            if (!rc && a[1] < 0) rc = fail(-8, "word %d: negative RMS delay", at);
            if (!rc) rc = gs_data(&S, a[0], 5 + 2ll * S.aw + (long long)a[1] * S.aw);
            if (!rc) ws_state(&S, a[0], 5 + 2ll * S.aw + (long long)a[1] * S.aw);
            break;
        case DSP_DCBLOCK:
            rc = gs_payload(&S, 1 + nf) || gs_data(&S, a[0], S.aw + 2); if (!rc) ws_state(&S, a[0], S.aw + 2); break;
        case DSP_DITHER:
            rc = gs_payload(&S, 1) || gs_data(&S, a[0], 3ll * S.aw); if (!rc) { ws_state(&S, a[0], 3ll * S.aw); ws_tpdf_user(&S); } break;
        case DSP_DITHER_NS2:
            rc = gs_payload(&S, 2) || gs_data(&S, a[0], 3) || gs_param(&S, a[1] + fi * 3, 3);
            if (!rc) { ws_state(&S, a[0], 3); ws_tpdf_user(&S); }
            break;
        case DSP_DISTRIB:
            rc = gs_payload(&S, 3);
            if (!rc && a[1] < 2) rc = fail(-8, "word %d: DISTRIB with %d bins", at, a[1]);
            if (!rc) rc = gs_io(&S, a[0], 1) || gs_data(&S, a[2], 1 + (long long)a[1]);
            if (!rc) { ws_write_io(&S, a[0]); ws_state(&S, a[2], 1 + (long long)a[1]); }
            break;
        case DSP_DIRAC: case DSP_SQUAREWAVE: case DSP_SINE:
            if (alu_int) {
                rc = fail(-8, "word %d: %s in int64 mode calls dspQNMmax()/DSP_Q31(), which the reference never "
                              "defines (dsp_runtime.c:1213-1305): no result to match", at, dspOpcodeText[op]);
                break;
            }
            rc = gs_payload(&S, 2 + nf) || gs_data(&S, a[0], op == DSP_SINE ? 2 * S.aw : 1);
            if (!rc) ws_state(&S, a[0], op == DSP_SINE ? 2 * S.aw : 1);
            break;
        default:
            rc = fail(-8, "word %d: unknown opcode %d", at, op);
        }
        if (rc) { free(smap); return g_err_code; }
        p += skip;
        if (G.next_skip_from && (int)(p - G.code) == G.next_skip_from) p = G.code + G.next_skip_to;      /* a piece that leaves a stretch out */
    }
    free(smap);
    d->skip_from = G.next_skip_from; d->skip_to = G.next_skip_from ? G.next_skip_to : 0;
    d->dither_only = nreal == 1 && S.w.tpdf_calc_seen;
    if (d->io_span == 0) d->io_span = 1;
    if (d->io_span > WAVE_IO_LIMIT) S.w.ok = S.w.complete = 0;
    if (S.w.ok) ws_finish(&S);
    d->wave_ok = S.w.ok;
    core_deps local;
    if (!deps) deps = &local;
    ws_export(&S, deps);
    /* for the device: what this core owns (written back after a launch; -1 = everything, the core runs alone) */
    d->tpdf_calc = deps->tpdf_calc;
    memcpy(d->early_io, deps->early_io, sizeof d->early_io);
    memcpy(d->written_io, deps->written_io, sizeof d->written_io);
    d->nown = -1;
    if (deps->complete) {
        static int32_t own[(DEPS_MAX_RANGES + 64) * 2];
        int n = 0;
        for (int i = 0; i < deps->nranges; i++) { own[2 * n] = deps->range[i][0]; own[2 * n + 1] = deps->range[i][1]; n++; }
        for (int i = 0; i < deps->nwr; i++) { own[2 * n] = deps->wr_word[i]; own[2 * n + 1] = deps->wr_word[i] + S.aw; n++; }
        d->nown = n; d->own = own;
    }
    return 0;
}

static int select_device(void)
{
    if (G.device_selected) {                            /* (another program's GPU, or somebody else's, may be the thread's current one) */
        if (avdsp_hip_set_device(G.device_ordinal)) return hip_fail();
        return 0;
    }
    int n = avdsp_hip_device_count();
    if (n <= 0) return fail(-10, "no HIP device: %s", avdsp_hip_last_error());
    int want = G.opt_device;
    if (want < 0) {
        const char *e = getenv("LOCAL_RANK");
        want = e ? atoi(e) % n : 0;
    }
    if (avdsp_hip_set_device(want)) return hip_fail();
    G.device_selected = 1;
    G.device_ordinal = want;
    return 0;
}

/* find or build the device plan of (core, format) */
/* end_word != 0: the plan of a strand group [core, end_word) of an interpreted core (dspRuntimeBlockAll) */
/* the program's device copy, made at the first call that needs it */
static int ensure_device(void)
{
    if (select_device()) return g_err_code;
    if (G.dev) return 0;
    G.dev = avdsp_hip_prog_create(G.total_words);
    if (!G.dev) return hip_fail();
    if (avdsp_hip_upload_words(G.dev, (const int32_t *)G.code, 0, G.total_words) ||
        avdsp_hip_tpdf_reset(G.dev, G.random, G.dither)) {
        hip_fail(); drop_device(); return g_err_code;
    }
    G.dev_state_valid = 1;
    avdsp_hip_profile_enable(G.dev, G.opt_profile);
    if (avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_PROFILE_STRIDE, G.opt_profile_stride > 0 ? G.opt_profile_stride : 1) ||
        avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_OVERLAP, G.opt_overlap) || avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_FIR_ROWS, G.opt_fir_rows) ||
        avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_READY_WORDS, G.opt_ready_words) || avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_LANE_HW, G.opt_lane_hw) || avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_FIR_SPLIT, G.opt_fir_split) || (G.opt_fir_launch_set && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_FIR_LAUNCH, G.opt_fir_launch)) || (G.opt_fir_lean_set && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_FIR_LEAN, G.opt_fir_lean)) || avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_RING_WAIT, G.opt_ring_wait) ||
        avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_HOST_SPLIT, G.opt_host_split) || avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_HOST_PIN, G.opt_host_pin) ||
        (G.opt_cu_split && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_CU_SPLIT, G.opt_cu_split)) ||
        (G.opt_group_serial && avdsp_hip_prog_set_option(G.dev, AVDSP_OPT_GROUP_FANOUT, 0)) ||
        (G.ninst > 1 && avdsp_hip_set_instances(G.dev, G.ninst))) {
        hip_fail(); drop_device(); return g_err_code;
    }
    return 0;
}

static core_plan *get_plan_range(int format, opcode_t *core, int end_word)
{
    if (!dspHeaderPtr || !G.code) { fail(-1, "no program loaded"); return 0; }
    if (!G.have_rate) { fail(-1, "dspRuntimeReset(fs) has not selected a sample rate yet"); return 0; }
    if (format < 2 || format > 6) { fail(-1, "DSP_FORMAT %d is not one of 2..6", format); return 0; }
    if (core < G.code || core >= G.code + dspHeaderPtr->totalLength) { fail(-1, "core pointer outside the loaded program"); return 0; }
    /* a program that runs as instances of chain cores has plans of ninst x its chains, which address ninst sample blocks: nothing
     * but dspRuntimeBlockAllInstancesDevice may launch them (this check sits in front of the plan cache on purpose) */
    if (G.chain_inst_made && !G.inst_call) {
        fail(-1, "this program runs as %d instances (dspRuntimeBlockAllInstancesDevice); dspRuntimeSetInstances(0) gives the single program back", G.ninst);
        return 0;
    }
    /* one key per core: a host may pass the DSP_CORE word or the first executable word behind it */
    if (!end_word) core = dspFindCoreBegin(core);
    for (int i = 0; i < G.nplans; i++)
        if (G.plans[i].core == core && G.plans[i].format == format && G.plans[i].end_word == end_word && G.plans[i].skip_from == G.next_skip_from) {
            if (G.plans[i].plan_id < 0 && !G.plans[i].empty) { fail(-8, "core was refused earlier"); return 0; }
            return &G.plans[i];
        }
    /* A chain plan owns the FIR histories of its core (device rings, written back into the mirror only when plans
     * are dropped).  The same core under a second DSP_FORMAT would get rings of its own that never see the
     * first's samples: go through the mirror instead -- drop every plan, lower again under the new format. */
    for (int i = 0; i < G.nplans; i++)
        if (G.plans[i].core == core && G.plans[i].format != format && !G.plans[i].end_word && !end_word &&
            (G.plans[i].max_taps || G.plans[i].nchains)) {
            if (replan()) return 0;
            break;
        }
    if (G.nplans == MAX_CORE_PLANS) { fail(-9, "too many cores"); return 0; }

    /* chains first (formats with parallel kernels); anything else the chain lowering calls "not
     * lowerable" (-8) is offered to the general interpreter; other failures are final */
    lowered L;
    avdsp_generic_desc gd;
    int chains = 0;
    memset(&L, 0, sizeof L);
    if (!G.opt_generic && !end_word) {              /* formats 2, 4, 6: pipelined cascade + MFMA FIR; 3, 5: one lane per chain */
        int rc = lower_core(format, core, &L);
        if (rc == 0) chains = 1;
        else { lowered_free(&L); if (rc != -8) return 0; }
    }
    core_deps deps;
    memset(&deps, 0, sizeof deps);                   /* chain cores: complete = 0, they run alone */
    if (!chains && scan_generic(format, core, end_word, &gd, &deps)) return 0;
    if (!chains) gd.tpdf_role = end_word ? G.next_tpdf_role : 0;

    if (ensure_device()) { lowered_free(&L); return 0; }
    core_plan *cp = &G.plans[G.nplans];
    cp->core = core; cp->format = format; cp->end_word = end_word; cp->skip_from = G.next_skip_from;
    cp->nchains = 0; cp->max_sections = 0; cp->max_taps = 0;
    cp->total_chains = 0; cp->first_chain = 0; cp->empty = 0;
    cp->deps = deps;
    if (chains) {
        /* this process's contiguous slice of the core's chains (dspRuntimeSetShard; the whole core when unsharded):
         * the plan, its FIR rings and the IO windows a block call must cover shrink to the slice */
        int lo, hi;
        shard_range(L.nchains, G.shard_world, G.shard_rank, &lo, &hi);
        cp->total_chains = L.nchains; cp->first_chain = lo; cp->nchains = hi - lo;
        if (hi == lo) {                                     /* more ranks than chains: nothing to do here */
            cp->empty = 1; cp->plan_id = -1;
            lowered_free(&L);
            G.nplans++;
            return cp;
        }
        const int sec0 = L.chains[lo].sec_base;
        const int sec1 = hi < L.nchains ? L.chains[hi].sec_base : L.nsec;
        for (int i = lo; i < hi; i++) L.chains[i].sec_base -= sec0;
        avdsp_plan_desc d;
        memset(&d, 0, sizeof d);
        d.format = format;
        d.nchains = hi - lo; d.chains = L.chains + lo;
        d.nsections = sec1 - sec0; d.sec_coef_word = L.coef_word + sec0; d.sec_state_word = L.state_word + sec0;
        d.store_mask = G.store_mask;
        /* Instances of a chain core (round 5): the chain list ninst times -- copy i of a chain reads and writes copy i of the mirror
         * (every word index + i * total words: its own state, FIR history and parameters) and block i of the callers' samples (IO
         * numbers + i * the blocks' distance: the kernels form `column = IO - window base`, which then lands in block i).  To the
         * kernels it is a bigger plan: 8 chains x 512 instances are the 4096 rows of one cascade launch. */
        avdsp_chain *xch = 0; int32_t *xco = 0, *xst = 0;
        if (G.chain_inst_made) {
            const int N = G.ninst, nc = hi - lo, ns = sec1 - sec0;
            xch = (avdsp_chain *)malloc(sizeof(avdsp_chain) * (size_t)N * (size_t)nc);
            xco = (int32_t *)malloc(sizeof(int32_t) * ((size_t)N * (size_t)ns + 1));
            xst = (int32_t *)malloc(sizeof(int32_t) * ((size_t)N * (size_t)ns + 1));
            if (!xch || !xco || !xst) { free(xch); free(xco); free(xst); lowered_free(&L); fail(-9, "out of memory"); return 0; }
            for (int i = 0; i < N; i++) {
                const long long wo = (long long)i * AVDSP_INSTANCE_STRIDE(G.total_words);
                for (int c = 0; c < nc; c++) {
                    avdsp_chain ch = L.chains[lo + c];
                    ch.in_io += (int)((long long)i * (long long)G.chain_inst_in);
                    for (int k = 0; k < ch.n_out; k++) ch.out_io[k] += (int)((long long)i * (long long)G.chain_inst_out);
                    ch.sec_base += i * ns;
                    if (ch.fir_taps) { ch.fir_coef_word += (int)wo; ch.fir_state_word += (int)wo; }
                    xch[(size_t)i * nc + c] = ch;
                }
                for (int j = 0; j < ns; j++) {
                    xco[(size_t)i * ns + j] = d.sec_coef_word[j] + (int)wo;
                    xst[(size_t)i * ns + j] = d.sec_state_word[j] + (int)wo;
                }
            }
            d.nchains = N * nc; d.chains = xch; d.nsections = N * ns; d.sec_coef_word = xco; d.sec_state_word = xst;
            d.instances = N;
        }
        cp->plan_id = avdsp_hip_prog_add_plan(G.dev, &d);
        free(xch); free(xco); free(xst);
        for (int i = lo; i < hi; i++) {
            if (L.chains[i].nsec > cp->max_sections) cp->max_sections = L.chains[i].nsec;
            if (L.chains[i].fir_taps > cp->max_taps) cp->max_taps = L.chains[i].fir_taps;
        }
        lowered_free(&L);
    } else {
        cp->plan_id = avdsp_hip_prog_add_generic(G.dev, &gd);
        /* a strand plan that cannot be attached (the device library has checks of its own, and an LDS request the device may refuse)
         * costs the lowering, not the call: the generic plan just made is complete, the stretch runs through the interpreter */
        if (cp->plan_id >= 0 && G.next_strands && end_word) (void)avdsp_hip_plan_add_strands(G.dev, cp->plan_id, G.next_strands);
    }
    if (cp->plan_id < 0) { hip_fail(); return 0; }
    G.nplans++;
    return cp;
}

static core_plan *get_plan(int format, opcode_t *core) { return get_plan_range(format, core, 0); }

int dspRuntimeKernelTime(int kind, double *total_ms, int *launches)
{
    if (total_ms) *total_ms = 0.0;
    if (launches) *launches = 0;
    if (!G.dev) return 0;
    device_current();
    if (avdsp_hip_profile_read(G.dev, kind, total_ms, launches)) return hip_fail();
    return 0;
}

/* Host-only: lowers the core (no device is touched) and reports what the plan would contain. */
int dspRuntimeCoreInfo(int format, opcode_t *core, int *nchains, int *max_sections, int *max_taps)
{
    (void)ctx_of(core);
    if (!dspHeaderPtr || !G.code) return fail(-1, "no program loaded");
    if (!G.have_rate) return fail(-1, "dspRuntimeReset(fs) has not selected a sample rate yet");
    if (format < 2 || format > 6) return fail(-1, "DSP_FORMAT %d is not one of 2..6", format);
    if (core < G.code || core >= G.code + dspHeaderPtr->totalLength) return fail(-1, "core pointer outside the loaded program");
    int nc = 0, ms = 0, mt = 0, rc = -8;
    if (!G.opt_generic) {
        lowered L;
        rc = lower_core(format, core, &L);
        if (rc == 0) {
            nc = L.nchains;
            for (int i = 0; i < L.nchains; i++) {
                if (L.chains[i].nsec > ms) ms = L.chains[i].nsec;
                if (L.chains[i].fir_taps > mt) mt = L.chains[i].fir_taps;
            }
        }
        lowered_free(&L);
    }
    if (rc == -8) {
        avdsp_generic_desc gd;
        rc = scan_generic(format, core, 0, &gd, 0);
    }
    if (rc) return rc;
    if (nchains) *nchains = nc;
    if (max_sections) *max_sections = ms;
    if (max_taps) *max_taps = mt;
    return 0;
}

/* Host-only: which chains of the core this process runs under the current dspRuntimeSetShard, and the IO numbers
 * they load and store -- what the block call's windows must cover, so a host can cut its column slice of a
 * [frames][channels] block for ANY loaded program.  total == 0: not a chain core (it runs whole on every rank). */
int dspRuntimeShardInfo(int format, opcode_t *core, int *total_chains, int *first_chain, int *nchains,
                        int *in_io_min, int *in_io_max, int *out_io_min, int *out_io_max)
{
    (void)ctx_of(core);
    if (!dspHeaderPtr || !G.code) return fail(-1, "no program loaded");
    if (!G.have_rate) return fail(-1, "dspRuntimeReset(fs) has not selected a sample rate yet");
    if (format < 2 || format > 6) return fail(-1, "DSP_FORMAT %d is not one of 2..6", format);
    if (core < G.code || core >= G.code + dspHeaderPtr->totalLength) return fail(-1, "core pointer outside the loaded program");
    int tot = 0, lo = 0, hi = 0, imin = 0, imax = -1, omin = 0, omax = -1;
    if (!G.opt_generic) {
        lowered L;
        int rc = lower_core(format, core, &L);
        if (rc == 0) {
            tot = L.nchains;
            shard_range(tot, G.shard_world, G.shard_rank, &lo, &hi);
            for (int i = lo; i < hi; i++) {
                const avdsp_chain *c = &L.chains[i];
                if (imax < imin || c->in_io < imin) imin = c->in_io;
                if (imax < imin || c->in_io > imax) imax = c->in_io;
                for (int k = 0; k < c->n_out; k++) {
                    if (omax < omin || c->out_io[k] < omin) omin = c->out_io[k];
                    if (omax < omin || c->out_io[k] > omax) omax = c->out_io[k];
                }
            }
        }
        lowered_free(&L);
        if (rc && rc != -8) return rc;
    }
    if (total_chains) *total_chains = tot;
    if (first_chain) *first_chain = lo;
    if (nchains) *nchains = hi - lo;
    if (in_io_min) *in_io_min = imin;
    if (in_io_max) *in_io_max = imax;
    if (out_io_min) *out_io_min = omin;
    if (out_io_max) *out_io_max = omax;
    return 0;
}

static int check_rundata(const int *rundata)
{
    if (rundata != (const int *)G.code + dspHeaderPtr->totalLength)
        return fail(-1, "rundata must be code + dspRuntimeInit() result (the state area of the loaded buffer)");
    return 0;
}

static int block_all(int format, int *rundata, const void *in, int in_stride, int in_io_base,
                     void *out, int out_stride, int out_io_base, int nframes, int on_device, void *stream, int pcm,
                     opcode_t *only);
/* an interpreted core goes to the device as its strand groups (block_all with that one core) */
static int takes_pieces(const core_plan *cp, int nframes, int in_stride, int in_io_base, int out_stride, int out_io_base)
{
    /* (windows that share IO numbers used to make every launch deliver whole rows, one after the other, and such a call kept its core in one
     * piece; since the shared columns are copied in front of a call's launches -- show_through, avdsp_kernels.hip -- they are like any others) */
    (void)in_stride; (void)in_io_base; (void)out_stride; (void)out_io_base;
    return cp->total_chains == 0 && G.opt_strand_split && nframes > 1;
}

int dspRuntimeBlockDevice(int format, opcode_t *core, int *rundata,
                          const void *d_in, int in_stride, int in_io_base,
                          void *d_out, int out_stride, int out_io_base, int nframes, void *stream)
{
    (void)ctx_of(core);
    core_plan *cp = get_plan(format, core);
    if (!cp) return g_err_code;
    if (check_rundata(rundata)) return -1;
    if (nframes <= 0 || cp->empty) return 0;
    if (avdsp_hip_wait_block_host(G.dev, 0) < 0) return hip_fail();       /* queued host blocks first */
    if (takes_pieces(cp, nframes, in_stride, in_io_base, out_stride, out_io_base))
        return block_all(format, rundata, d_in, in_stride, in_io_base, d_out, out_stride, out_io_base, nframes, 1, stream, AVDSP_PCM_S32, core);
    if (avdsp_hip_run_block(G.dev, cp->plan_id, d_in, in_stride, in_io_base, d_out, out_stride, out_io_base,
                            nframes, G.opt_fir_impl, G.opt_biquad_impl, stream))
        return hip_fail();
    return 0;
}

static int block_host(int format, opcode_t *core, int *rundata, const void *in, int in_stride, int in_io_base,
                      void *out, int out_stride, int out_io_base, int nframes)
{
    (void)ctx_of(core);
    core_plan *cp = get_plan(format, core);
    if (!cp) return g_err_code;
    if (check_rundata(rundata)) return -1;
    if (nframes <= 0 || cp->empty) return 0;
    if (avdsp_hip_wait_block_host(G.dev, 0) < 0) return hip_fail();       /* queued host blocks first */
    if (takes_pieces(cp, nframes, in_stride, in_io_base, out_stride, out_io_base))
        return block_all(format, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes, 0, 0, AVDSP_PCM_S32, core);
    if (avdsp_hip_run_block_host(G.dev, cp->plan_id, in, in_stride, in_io_base, out, out_stride, out_io_base,
                                 nframes, G.opt_fir_impl, G.opt_biquad_impl))
        return hip_fail();
    return 0;
}

/* dspRuntimeBlock_N as a queue (linux/avdsp_plugin.c:98-141 with the next period already on its way): returns the number
 * of blocks in flight, < 0 on error.  The buffers stay the library's until dspRuntimeBlockWait lets the block through. */
int dspRuntimeBlockSubmit(int format, opcode_t *core, int *rundata, const void *in, int in_stride, int in_io_base,
                          void *out, int out_stride, int out_io_base, int nframes)
{
    (void)ctx_of(core);
    core_plan *cp = get_plan(format, core);
    if (!cp) return g_err_code;
    if (check_rundata(rundata)) return -1;
    if (nframes <= 0 || cp->empty) return avdsp_hip_wait_block_host(G.dev, 1 << 30);
    if (takes_pieces(cp, nframes, in_stride, in_io_base, out_stride, out_io_base)) {
        if (avdsp_hip_wait_block_host(G.dev, 0) < 0) return hip_fail();
        return block_all(format, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes, 0, 0, AVDSP_PCM_S32, core);
    }
    const int rc = avdsp_hip_submit_block_host(G.dev, cp->plan_id, in, in_stride, in_io_base, out, out_stride, out_io_base,
                                               nframes, G.opt_fir_impl, G.opt_biquad_impl);
    if (rc < 0) return hip_fail();
    return rc;
}

int dspRuntimeBlockWait(int max_in_flight)
{
    if (!G.dev) return 0;
    device_current();
    const int rc = avdsp_hip_wait_block_host(G.dev, max_in_flight);
    if (rc < 0) return hip_fail();
    return rc;
}

int dspRuntimeBlock_2(opcode_t *core, int *rundata, const int *in, int in_stride, int in_io_base,
                      int *out, int out_stride, int out_io_base, int nframes)
{ return block_host(2, core, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes); }

int dspRuntimeBlock_3(opcode_t *core, int *rundata, const int *in, int in_stride, int in_io_base,
                      int *out, int out_stride, int out_io_base, int nframes)
{ return block_host(3, core, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes); }

int dspRuntimeBlock_5(opcode_t *core, int *rundata, const float *in, int in_stride, int in_io_base,
                      float *out, int out_stride, int out_io_base, int nframes)
{ return block_host(5, core, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes); }

int dspRuntimeBlock_4(opcode_t *core, int *rundata, const int *in, int in_stride, int in_io_base,
                      int *out, int out_stride, int out_io_base, int nframes)
{ return block_host(4, core, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes); }

int dspRuntimeBlock_6(opcode_t *core, int *rundata, const float *in, int in_stride, int in_io_base,
                      float *out, int out_stride, int out_io_base, int nframes)
{ return block_host(6, core, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes); }

/* ---- all cores of the program over one block ----
 * Same result as dspRuntimeBlock_N for core 1, 2, ... in turn (the host loop of linux/avdsp_plugin.c:95-142), in one
 * call: samples cross PCIe once, and cores that do not meet run at the same time.  Two cores meet when one stores a
 * frame slot, a memory word (STORE_MEM, LOAD_MUX / TPDF result), a state range or the dither globals that the other
 * reads before writing it, or writes too; scan_generic collected that per core.  Levels: a core goes one level
 * behind the latest earlier core it meets; the cores of a level are handed to the device together.            */
static int cores_meet_ex(const core_deps *a, const core_deps *b, int ignore_tpdf)
{
    if (!a->complete || !b->complete) return 1;
    for (int k = 0; k < 8; k++)
        if ((a->written_io[k] & (b->early_io[k] | b->written_io[k])) || (a->early_io[k] & b->written_io[k])) return 1;
    for (int i = 0; i < a->nwr; i++) {
        for (int j = 0; j < b->nrd; j++) { int d = a->wr_word[i] - b->rd_word[j]; if (d > -2 && d < 2) return 1; }
        for (int j = 0; j < b->nwr; j++) { int d = a->wr_word[i] - b->wr_word[j]; if (d > -2 && d < 2) return 1; }
        for (int j = 0; j < b->nranges; j++) if (a->wr_word[i] + 2 > b->range[j][0] && a->wr_word[i] < b->range[j][1]) return 1;
    }
    for (int i = 0; i < a->nwr; i++)
        for (int j = 0; j < b->nparams; j++) if (a->wr_word[i] + 2 > b->param[j][0] && a->wr_word[i] < b->param[j][1]) return 1;
    for (int i = 0; i < b->nwr; i++)
        for (int j = 0; j < a->nparams; j++) if (b->wr_word[i] + 2 > a->param[j][0] && b->wr_word[i] < a->param[j][1]) return 1;
    for (int i = 0; i < a->nrd; i++)
        for (int j = 0; j < b->nwr; j++) { int d = a->rd_word[i] - b->wr_word[j]; if (d > -2 && d < 2) return 1; }
    for (int i = 0; i < b->nwr; i++)
        for (int j = 0; j < a->nranges; j++) if (b->wr_word[i] + 2 > a->range[j][0] && b->wr_word[i] < a->range[j][1]) return 1;
    for (int i = 0; i < a->nranges; i++)
        for (int j = 0; j < b->nranges; j++)
            if (a->range[i][0] < b->range[j][1] && b->range[j][0] < a->range[i][1]) return 1;
    /* a data word one side reads (LOAD_MEM / LOAD_MEM_DATA, two words) inside a state range the other side's opcodes keep rewriting */
    for (int i = 0; i < a->nrd; i++)
        for (int j = 0; j < b->nranges; j++) if (a->rd_word[i] + 2 > b->range[j][0] && a->rd_word[i] < b->range[j][1]) return 1;
    for (int i = 0; i < b->nrd; i++)
        for (int j = 0; j < a->nranges; j++) if (b->rd_word[i] + 2 > a->range[j][0] && b->rd_word[i] < a->range[j][1]) return 1;
    if (!ignore_tpdf && ((a->tpdf_calc && (b->tpdf_user || b->tpdf_calc)) || (a->tpdf_user && b->tpdf_calc))) return 1;
    return 0;
}
static int cores_meet(const core_deps *a, const core_deps *b) { return cores_meet_ex(a, b, 0); }

/* ---- strand groups ----
 * Inside a core the strands -- LOAD ... STORE runs -- mostly do not depend on each other either: a strand that
 * begins by replacing X and never looks at the Y it inherits carries nothing over from the strand in front of it.
 * Cut there, a core becomes several pieces that go through the same "do they meet" test as whole cores and may run
 * at the same time (a wide program with one strand per channel turns into many workgroups instead of one wave).
 * A cut is legal in front of LOAD / LOAD_GAIN / LOAD_MEM / LOAD_MUX / CLRXY when, from there on, nothing reads Y
 * before Y has been set from this piece's own X; a core with a DSP_TPDF opcode (which switches the dither width
 * for the rest of the frame) is left whole.  Returns the number of pieces, starts[] = their first opcode words. */
#define MAX_GROUPS 64
static int reads_y(int op)
{
    switch (op) {
    case DSP_SWAPXY: case DSP_COPYYX: case DSP_ADDXY: case DSP_ADDYX: case DSP_SUBXY: case DSP_SUBYX: case DSP_MULXY:
    case DSP_DIVXY: case DSP_DIVYX: case DSP_AVGXY: case DSP_AVGYX: case DSP_NEGY: case DSP_RMS:
        return 1;
    }
    return 0;
}
static int sets_y_from_x(int op)
{
    switch (op) {
    case DSP_LOAD: case DSP_LOAD_GAIN: case DSP_LOAD_MEM: case DSP_VALUE: case DSP_VALUE_INT: case DSP_DELAY_1:
    case DSP_COPYXY: case DSP_CLRXY: case DSP_SINE:
        return 1;
    }
    return 0;
}
static int split_core(opcode_t *core, int *starts)
{
    const int total = dspHeaderPtr->totalLength;
    opcode_t *p0 = dspFindCoreBegin(core);
    enum { MAXOPS = 32768 };
    static int at[MAXOPS], op[MAXOPS], cut[MAXOPS];
    int n = 0;
    for (opcode_t *p = p0;;) {
        const int w = (int)(p - G.code);
        if (w < 0 || w >= total) return 0;                     /* damaged: the scan will say so */
        const int o = p->op.opcode; const unsigned skip = p->op.skip;
        if (skip == 0 || o == DSP_CORE || o == DSP_END_OF_CODE) break;
        if (o == DSP_TPDF) { n = -1; break; }
        if (n == MAXOPS) { n = -1; break; }
        at[n] = w; op[n] = o; n++;
        p += skip;
    }
    starts[0] = (int)(p0 - G.code);
    if (n < 4) return 1;
    /* legal cuts */
    int ncut = 0;
    for (int i = 1; i < n; i++) {
        if (op[i] != DSP_LOAD && op[i] != DSP_LOAD_GAIN && op[i] != DSP_LOAD_MEM && op[i] != DSP_LOAD_MUX && op[i] != DSP_CLRXY) continue;
        int legal = 1;
        if (op[i] != DSP_CLRXY)
            for (int k = i + 1; k < n; k++) {
                if (reads_y(op[k])) { legal = 0; break; }
                if (sets_y_from_x(op[k])) break;
            }
        if (legal) cut[ncut++] = i;
    }
    if (!ncut) return 1;
    /* Pieces worth a wave of their own, at most MAX_GROUPS of them, as even as the cuts allow.  What a piece costs is its
     * opcodes whose state runs from frame to frame (a cascade or a FIR is 100 ns per frame, a gain or a store 4: tools/interp_opcost.py),
     * so those count four: a strand of LOAD_GAIN, BIQUADS, STORE_MEM is a piece (dacdiy1.bin's first core: two of them, 0.23 -> 0.13 us). */
    static int wsum[MAXOPS + 1];
    wsum[0] = 0;
    for (int i = 0; i < n; i++) {
        int w = 1;
        switch (op[i]) {
        case DSP_BIQUADS: case DSP_FIR: case DSP_RMS: case DSP_DCBLOCK: case DSP_DITHER: case DSP_DITHER_NS2:
        case DSP_DISTRIB: case DSP_SINE: case DSP_TPDF_CALC: w = 4; break;
        }
        wsum[i + 1] = wsum[i] + w;
    }
#define PIECE_W(lo, hi) (wsum[hi] - wsum[lo])
    if (wsum[n] < 8) return 1;
    int groups = wsum[n] / 4;
    if (groups > MAX_GROUPS) groups = MAX_GROUPS;
    if (groups > ncut + 1) groups = ncut + 1;
    int ng = 1, last = 0;
    /* the piece with the TPDF_CALC is a level of its own in front of every piece that dithers: keep it as short as the
     * cuts allow (the first legal cut behind the TPDF_CALC), whatever its size */
    for (int i = 0; i < n; i++)
        if (op[i] == DSP_TPDF_CALC) {
            for (int c = 0; c < ncut; c++)
                if (cut[c] > i) { if (PIECE_W(cut[c], n) >= 4) { starts[ng++] = at[cut[c]]; last = cut[c]; } break; }
            break;
        }
    for (int g = 1; g < groups && ng < MAX_GROUPS; g++) {
        const int want = (int)((long long)wsum[n] * g / groups);
        int best = -1;
        for (int c = 0; c < ncut; c++) if (wsum[cut[c]] >= want && cut[c] > last) { best = cut[c]; break; }
        if (best < 0 || PIECE_W(best, n) < 4 || PIECE_W(last, best) < 4) continue;
        starts[ng++] = at[best];
        last = best;
    }
#undef PIECE_W
    return ng;
}

/* ---- strand runs on lanes (include/avdsp_hip.h, strand plans) ----
 * A stretch [start, end) of an interpreted core is offered as a strand plan when it is N >= 2 repetitions of one opcode sequence
 * -- every repetition opening at a legal cut (split_core's rule) with LOAD / LOAD_GAIN / LOAD_MEM, made of opcodes the strand
 * kernel has, equal in every word that steers control (opcode, length, sections per bank) -- and the repetitions are strangers:
 * no IO stored twice or stored by one and loaded by another (or by itself earlier in the frame), state areas disjoint, no memory
 * word written by one that any of them reads.  scan_generic has bounds-checked every offset of the stretch before this runs.  */
typedef struct { long long lo, hi; int kind; } s_interval;           /* kind 0: state (exclusive), 1: word read, 2: word written */
typedef struct { avdsp_strand_op *ops; int32_t *args; avdsp_strand_desc d;
                 s_interval *iv; int niv; int *stored, nst, *loaded, nld; } strand_plan;        /* (what the run touches: against a prefix piece) */
static void strand_free(strand_plan *S) { free(S->ops); free(S->args); free(S->iv); free(S->stored); free(S->loaded); memset(S, 0, sizeof *S); }
static int cmp_interval(const void *a, const void *b) { const s_interval *x = a, *y = b; return (x->lo > y->lo) - (x->lo < y->lo); }

static int sop_of(int op)
{
    switch (op) {
    case DSP_LOAD: return AVDSP_SOP_LOAD;               case DSP_LOAD_GAIN: return AVDSP_SOP_LOAD_GAIN;
    case DSP_GAIN: return AVDSP_SOP_GAIN;               case DSP_COPYXY: return AVDSP_SOP_COPYXY;
    case DSP_SWAPXY: return AVDSP_SOP_SWAPXY;           case DSP_COPYYX: return AVDSP_SOP_COPYYX;
    case DSP_ADDXY: return AVDSP_SOP_ADDXY;             case DSP_ADDYX: return AVDSP_SOP_ADDYX;
    case DSP_SUBXY: return AVDSP_SOP_SUBXY;             case DSP_SUBYX: return AVDSP_SOP_SUBYX;
    case DSP_NEGX: return AVDSP_SOP_NEGX;               case DSP_SHIFT: return AVDSP_SOP_SHIFT;
    case DSP_SAT0DB: return AVDSP_SOP_SAT0DB;           case DSP_SAT0DB_TPDF: return AVDSP_SOP_SAT0DB_TPDF;
    case DSP_SAT0DB_GAIN: return AVDSP_SOP_SAT0DB_GAIN; case DSP_SAT0DB_TPDF_GAIN: return AVDSP_SOP_SAT0DB_TPDF_GAIN;
    case DSP_STORE: return AVDSP_SOP_STORE;             case DSP_LOAD_MEM: return AVDSP_SOP_LOAD_MEM;
    case DSP_STORE_MEM: return AVDSP_SOP_STORE_MEM;     case DSP_DELAY: return AVDSP_SOP_DELAY;
    case DSP_DELAY_DP: return AVDSP_SOP_DELAY_DP;       case DSP_BIQUADS: return AVDSP_SOP_BIQUADS;
    }
    return 0;
}

/* 0 = the stretch is a strand run (S filled), -8 = it is not (no error text: the caller falls back), -9 = out of memory */
static int strand_lower(int format, int start, int end, strand_plan *S)
{
    memset(S, 0, sizeof *S);
    const int prog_words = dspHeaderPtr->totalLength, aw = (format == 3 || format == 5) ? 1 : 2;
    enum { MAXOPS = 1 << 20 };
    int nops_all = 0, cap = 4096;
    int *at = (int *)malloc((size_t)cap * sizeof(int));
    if (!at) return -9;
    for (int w = start; w < end;) {
        const opcode_t *p = G.code + w;
        const int op = p->op.opcode; const unsigned skip = p->op.skip;
        if (skip == 0 || op == DSP_CORE || op == DSP_END_OF_CODE) break;
        if (op != DSP_NOP && op != DSP_PARAM && op != DSP_PARAM_NUM && op != DSP_SERIAL) {
            if (!sop_of(op) || nops_all == MAXOPS) { free(at); return -8; }
            if (nops_all == cap) { cap *= 2; int *q = (int *)realloc(at, (size_t)cap * sizeof(int)); if (!q) { free(at); return -9; } at = q; }
            at[nops_all++] = w;
        }
        w += (int)skip;
    }
    if (nops_all < 4) { free(at); return -8; }
#define OPC(i) (G.code[at[i]].op.opcode)
    /* the first repetition: from the first opcode to the next legal cut whose opcode equals the first one's; then the stretch must
     * be whole repetitions of that length */
    const int op0 = OPC(0);
    if (op0 != DSP_LOAD && op0 != DSP_LOAD_GAIN && op0 != DSP_LOAD_MEM) { free(at); return -8; }
    int len = 0;
    for (int i = 1; i < nops_all && !len; i++) {
        if (OPC(i) != op0) continue;
        int legal = 1;
        for (int k = i + 1; k < nops_all; k++) {
            if (reads_y(OPC(k))) { legal = 0; break; }
            if (sets_y_from_x(OPC(k))) break;
        }
        if (!legal) continue;
        /* a candidate period: the stretch must repeat with it */
        if (nops_all % i) continue;
        int same = 1;
        for (int k = i; k < nops_all && same; k++)
            same = OPC(k) == OPC(k - i) && G.code[at[k]].op.skip == G.code[at[k - i]].op.skip;
        if (same) len = i;
    }
    if (!len) { free(at); return -8; }
    const int N = nops_all / len;
    if (N < 2) { free(at); return -8; }
    /* every repetition opens at a legal cut */
    for (int r = 1; r < N; r++) {
        const int i = r * len;
        for (int k = i + 1; k < nops_all; k++) {
            if (reads_y(OPC(k))) { free(at); return -8; }
            if (sets_y_from_x(OPC(k))) break;
        }
    }
    S->ops = (avdsp_strand_op *)calloc((size_t)len, sizeof *S->ops);
    if (!S->ops) { free(at); return -9; }
    int nargs = 0;
    for (int j = 0; j < len; j++) {
        avdsp_strand_op *o = &S->ops[j];
        o->op = sop_of(OPC(j));
        switch (o->op) {
        case AVDSP_SOP_LOAD: case AVDSP_SOP_GAIN: case AVDSP_SOP_SHIFT: case AVDSP_SOP_SAT0DB_GAIN: case AVDSP_SOP_SAT0DB_TPDF_GAIN:
        case AVDSP_SOP_STORE: case AVDSP_SOP_LOAD_MEM: case AVDSP_SOP_STORE_MEM:
            o->a0 = nargs++; break;
        case AVDSP_SOP_LOAD_GAIN: case AVDSP_SOP_BIQUADS:
            o->a0 = nargs++; o->a1 = nargs++; break;
        case AVDSP_SOP_DELAY: case AVDSP_SOP_DELAY_DP:
            o->a0 = nargs++; o->a1 = nargs++; o->a2 = nargs++; break;
        default: break;
        }
    }
    if (nargs == 0) nargs = 1;
    S->args = (int32_t *)calloc((size_t)N * nargs, sizeof(int32_t));
    s_interval *iv = (s_interval *)malloc(((size_t)N * len * 2 + 4) * sizeof *iv);
    int *stored = (int *)malloc(((size_t)N * len + 1) * sizeof(int)), *loaded = (int *)malloc(((size_t)N * len + 1) * sizeof(int));
    int rc = 0, niv = 0, nst = 0, nld = 0, io_max = 0;
    if (!S->args || !iv || !stored || !loaded) rc = -9;
    for (int r = 0; r < N && !rc; r++) {
        int32_t *row = S->args + (size_t)r * nargs;
        const int st0 = nst;                                 /* this strand's own stores so far */
        for (int j = 0; j < len && !rc; j++) {
            const opcode_t *p = G.code + at[r * len + j];
            const int *w = (const int *)p + 1;
            const int here = at[r * len + j];
            const avdsp_strand_op *o = &S->ops[j];
#define SL_WORD(off, n, kind_) do { const long long lo_ = (long long)here + (off); if (lo_ < 12 || lo_ + (n) > prog_words) rc = -8; \
                                    else { iv[niv].lo = lo_; iv[niv].hi = lo_ + (n); iv[niv].kind = (kind_); niv++; } } while (0)
            switch (o->op) {
            case AVDSP_SOP_LOAD: case AVDSP_SOP_LOAD_GAIN:
                if (w[0] < 0 || w[0] >= GENERIC_IO_LIMIT) { rc = -8; break; }
                for (int k = st0; k < nst; k++) if (stored[k] == w[0]) rc = -8;      /* the frame's own value: a slot, not the block */
                row[o->a0] = w[0]; loaded[nld++] = w[0];
                if (o->op == AVDSP_SOP_LOAD_GAIN) { SL_WORD(w[1], 1, 1); row[o->a1] = here + w[1]; }
                break;
            case AVDSP_SOP_GAIN: case AVDSP_SOP_SAT0DB_GAIN: case AVDSP_SOP_SAT0DB_TPDF_GAIN:
                SL_WORD(w[0], 1, 1); row[o->a0] = here + w[0];
                break;
            case AVDSP_SOP_SHIFT: row[o->a0] = w[0]; break;
            case AVDSP_SOP_STORE:
                if (w[0] < 0 || w[0] >= GENERIC_IO_LIMIT) { rc = -8; break; }
                row[o->a0] = w[0]; stored[nst++] = w[0];
                if (w[0] > io_max) io_max = w[0];
                break;
            case AVDSP_SOP_LOAD_MEM:  SL_WORD(w[0], aw, 1); row[o->a0] = here + w[0]; break;
            case AVDSP_SOP_STORE_MEM: SL_WORD(w[0], aw, 2); row[o->a0] = here + w[0]; break;
            case AVDSP_SOP_DELAY: case AVDSP_SOP_DELAY_DP: {
                /* :769-790: with a parameter the first word is the line's size in samples, without it microseconds (scan_generic) */
                const unsigned dfac = (unsigned)(4294.967296 * (double)dspConvertFrequencyFromIndex(G.fs_index));
                const long long nline = w[2] ? (long long)w[0] : (long long)(((unsigned long long)(unsigned)w[0] * dfac) >> 32);
                const long long words = 1 + nline * (o->op == AVDSP_SOP_DELAY_DP ? aw : 1);
                if (w[0] < 0) { rc = -8; break; }
                if (w[1] < 0 || w[1] + words > dspHeaderPtr->dataSize) { rc = -8; break; }
                row[o->a0] = w[0]; row[o->a1] = prog_words + w[1];
                row[o->a2] = 0;
                if (w[2]) { SL_WORD(w[2], 1, 1); row[o->a2] = here + w[2]; }
                iv[niv].lo = (long long)prog_words + w[1]; iv[niv].hi = iv[niv].lo + words; iv[niv].kind = 0; niv++;
                break; }
            case AVDSP_SOP_BIQUADS: {
                const long long bw = (long long)here + w[1];
                if (bw < 12 || bw + 2 > prog_words) { rc = -8; break; }
                const int num = (short)G.code[bw].i32;
                if (num < 1 || num > 64) { rc = -8; break; }
                if (r == 0) S->ops[j].imm = num; else if (S->ops[j].imm != num) { rc = -8; break; }
                if (w[0] < 0 || (long long)w[0] + 6ll * num > dspHeaderPtr->dataSize) { rc = -8; break; }
                SL_WORD(w[1], G.biquad_offset + (num - 1) * dspBiquadFreqSkip + 5, 1);
                row[o->a0] = prog_words + w[0]; row[o->a1] = (int)bw;
                iv[niv].lo = (long long)prog_words + w[0]; iv[niv].hi = iv[niv].lo + 6ll * num; iv[niv].kind = 0; niv++;
                break; }
            default: break;
            }
#undef SL_WORD
        }
    }
#undef OPC
    if (!rc) {
        /* no IO stored twice; none stored by one strand and loaded by any */
        qsort(stored, (size_t)nst, sizeof(int), cmp_int);
        for (int i = 1; i < nst && !rc; i++) if (stored[i] == stored[i - 1]) rc = -8;
        for (int i = 0; i < nld && !rc; i++) if (bsearch(&loaded[i], stored, (size_t)nst, sizeof(int), cmp_int)) rc = -8;
        /* state areas disjoint from everything; a written word disjoint from every word read or written */
        qsort(iv, (size_t)niv, sizeof *iv, cmp_interval);
        long long wr_reach = -1, st_reach = -1, rd_reach = -1;      /* furthest end seen per kind (sorted by start) */
        for (int i = 0; i < niv && !rc; i++) {
            const s_interval *v = &iv[i];
            if (v->kind == 0) { if (v->lo < st_reach || v->lo < wr_reach || v->lo < rd_reach) rc = -8; if (v->hi > st_reach) st_reach = v->hi; }
            else if (v->kind == 2) { if (v->lo < st_reach || v->lo < wr_reach || v->lo < rd_reach) rc = -8; if (v->hi > wr_reach) wr_reach = v->hi; }
            else { if (v->lo < st_reach || v->lo < wr_reach) rc = -8; if (v->hi > rd_reach) rd_reach = v->hi; }
        }
    }
    free(at);
    S->iv = iv; S->niv = niv; S->stored = stored; S->nst = nst; S->loaded = loaded; S->nld = nld;
    if (rc) { strand_free(S); return rc; }
    int nres = 0;
    for (int j = 0; j < len; j++) { S->ops[j].rcol = nres; nres += avdsp_strand_rcols(S->ops[j].op, S->ops[j].imm, aw); }
    if (nres > 224 || len > 512) { strand_free(S); return -8; }   /* the kernel's table: 64 lanes x nres words of LDS (56 KB); its operation list: 512 */
    S->d.nops = len; S->d.ops = S->ops; S->d.nstrands = N; S->d.nargs = nargs; S->d.args = S->args; S->d.stored_io_max = io_max; S->d.nres = nres;
    return 0;
}

/* does the piece in front of a run (its deps from scan_generic) meet the run?  cores_meet_ex on what strand_lower collected (a run
 * of thousands of strands does not fit a core_deps) */
static int prefix_meets_run(const core_deps *d0, const strand_plan *S)
{
    if (!d0->complete) return 1;
    for (int i = 0; i < S->nst; i++) {
        const int io = S->stored[i];
        if (io < 256 && ((d0->written_io[io >> 5] | d0->early_io[io >> 5]) >> (io & 31) & 1u)) return 1;
    }
    for (int i = 0; i < S->nld; i++) {
        const int io = S->loaded[i];
        if (io < 256 && (d0->written_io[io >> 5] >> (io & 31) & 1u)) return 1;
    }
    for (int i = 0; i < S->niv; i++) {
        const s_interval *v = &S->iv[i];
        for (int j = 0; j < d0->nwr; j++) if (d0->wr_word[j] + 2 > v->lo && d0->wr_word[j] < v->hi) return 1;
        for (int j = 0; j < d0->nranges; j++) if (d0->range[j][1] > v->lo && d0->range[j][0] < v->hi) return 1;
        if (v->kind != 1) for (int j = 0; j < d0->nrd; j++) if (d0->rd_word[j] + 2 > v->lo && d0->rd_word[j] < v->hi) return 1;
        if (v->kind == 2) for (int j = 0; j < d0->nparams; j++) if (d0->param[j][1] > v->lo && d0->param[j][0] < v->hi) return 1;
    }
    return 0;
}

/* the first LOAD / LOAD_GAIN / LOAD_MEM of the core from which everything up to the core's end is one strand run; 0 = found */
static int find_strand_run(int format, opcode_t *begin, int *run_word, int *end_word, strand_plan *S)
{
    const int total = dspHeaderPtr->totalLength;
    int core_end = -1, tried = 0;
    for (opcode_t *p = begin;;) {
        const int w = (int)(p - G.code);
        if (w < 0 || w >= total) return -8;
        if (p->op.skip == 0 || p->op.opcode == DSP_CORE || p->op.opcode == DSP_END_OF_CODE) { core_end = w; break; }
        if (p->op.opcode == DSP_TPDF) return -8;           /* switches the dither width for the rest of the frame: the core stays whole */
        p += p->op.skip;
    }
    for (opcode_t *p = begin; tried < 12;) {
        const int w = (int)(p - G.code);
        if (w >= core_end) break;
        const int o = p->op.opcode;
        if (o == DSP_LOAD || o == DSP_LOAD_GAIN || o == DSP_LOAD_MEM) {
            tried++;
            const int rc = strand_lower(format, w, core_end, S);
            if (rc == -9) return -9;
            if (rc == 0) { *run_word = w; *end_word = core_end; return 0; }
        }
        p += p->op.skip;
    }
    return -8;
}

/* A wave at lane = strand costs the same for 2 strands as for 64 (0.39 us per frame for a strand of gains, a cascade, a delay line and
 * a dithered store), and a strand's cheap opcodes cost it ten times what they cost the frame-parallel interpreter, which gives up to
 * 64 strand groups a wave each: 16 such strands 0.23 us per frame there against 0.46 on lanes, 100 strands (two per group) 0.52
 * against 0.45 (tools/wide_core_bench.py).  So: runs of more than 64 strands; "strand_lanes" 2 lowers every run there is. */
static int strand_lanes_from(void) { return G.opt_strand_lanes >= 2 ? 2 : 65; }

/* Host-only: would the core's tail run as a strand plan?  strands = 0: no.  (prefix_words: opcode words in front of the run that
 * stay with the interpreter; ops: micro-operations per strand) */
int dspRuntimeStrandInfo(int format, opcode_t *core, int *strands, int *ops, int *prefix_words)
{
    (void)ctx_of(core);
    if (!dspHeaderPtr || !G.code) return fail(-1, "no program loaded");
    if (!G.have_rate) return fail(-1, "dspRuntimeReset(fs) has not selected a sample rate yet");
    if (format < 2 || format > 6) return fail(-1, "DSP_FORMAT %d is not one of 2..6", format);
    if (core < G.code || core >= G.code + dspHeaderPtr->totalLength) return fail(-1, "core pointer outside the loaded program");
    if (ensure_encoding(format)) return g_err_code;
    opcode_t *begin = dspFindCoreBegin(core);
    strand_plan S;
    int w = 0, e = 0;
    int rc = find_strand_run(format, begin, &w, &e, &S);
    if (rc == -9) return fail(-9, "out of memory");
    if (rc == 0 && (!G.opt_strand_lanes || S.d.nstrands < strand_lanes_from())) { strand_free(&S); rc = -8; }
    if (strands) *strands = rc ? 0 : S.d.nstrands;
    if (ops) *ops = rc ? 0 : S.d.nops;
    if (prefix_words) *prefix_words = rc ? 0 : w - (int)(begin - G.code);
    if (!rc) strand_free(&S);
    return 0;
}

/* Is the stretch [start, end) one strand of the shape  LOAD | LOAD_GAIN | LOAD_MEM, COPYXY, <way 1>, SWAPXY, <way 2>  with both ways made of
 * opcodes that work on X alone (gains, cascades, FIRs, delay lines, saturation, stores) and each holding a cascade or a FIR (else the
 * fork is not worth a wave)?  Then way 2 sees exactly the load's value (the copy in Y, untouched by way 1) and nothing of way 1, and the
 * Y it leaves behind is never read (the strand ends the piece, and a piece ends at a legal cut).  Words of the COPYXY, the SWAPXY and
 * the opcode behind it are returned. */
static int two_way_fork(int start, int end, int *copy_word, int *swap_word, int *after_word)
{
    int i = 0, heavy1 = 0, heavy2 = 0, seen_swap = 0;
    for (int w = start; w < end;) {
        const opcode_t *p = G.code + w;
        const int op = p->op.opcode; const unsigned skip = p->op.skip;
        if (skip == 0 || op == DSP_CORE || op == DSP_END_OF_CODE) break;
        if (op != DSP_NOP && op != DSP_PARAM && op != DSP_PARAM_NUM && op != DSP_SERIAL) {
            if (i == 0) { if (op != DSP_LOAD && op != DSP_LOAD_GAIN && op != DSP_LOAD_MEM) return 0; }
            else if (i == 1) { if (op != DSP_COPYXY) return 0; *copy_word = w; }
            else if (op == DSP_SWAPXY) {
                if (seen_swap) return 0;
                seen_swap = 1; *swap_word = w; *after_word = w + (int)skip;
            } else {
                switch (op) {
                case DSP_BIQUADS: case DSP_FIR: if (seen_swap) heavy2 = 1; else heavy1 = 1; break;
                case DSP_GAIN: case DSP_DELAY: case DSP_DELAY_DP: case DSP_SHIFT: case DSP_NEGX: case DSP_STORE:
                case DSP_SAT0DB: case DSP_SAT0DB_TPDF: case DSP_SAT0DB_GAIN: case DSP_SAT0DB_TPDF_GAIN: break;
                default: return 0;
                }
            }
            i++;
        }
        w += (int)skip;
    }
    return seen_swap && heavy1 && heavy2 && *after_word < end;
}

/* one core -> its pieces (or itself), appended to cp[] */
static int expand_core(int format, opcode_t *c, core_plan **cp, int *pn)
{
    int n = *pn;
    opcode_t *begin = dspFindCoreBegin(c);
    int starts[MAX_GROUPS + 1], ng = 1;
    static core_deps pd[MAX_GROUPS];
    core_plan *whole = get_plan(format, begin);           /* also tells whether the core is a chain core */
    if (!whole) return g_err_code;
    /* [prefix] + a run of identical strands up to the core's end: the run goes to lanes (strand_lower), the prefix -- usually the
     * TPDF_CALC -- stays a piece of the interpreter's, one level ahead */
    if (G.opt_strand_split && G.opt_strand_lanes && whole->total_chains == 0) {
        strand_plan S;
        int w = 0, core_end = 0;
        int rc = find_strand_run(format, begin, &w, &core_end, &S);
        if (rc == -9) return fail(-9, "out of memory");
        if (rc == 0 && S.d.nstrands < strand_lanes_from()) { strand_free(&S); rc = -8; }
        if (rc == 0) {
            avdsp_generic_desc gd;
            static core_deps d0, d1;
            int ok = !scan_generic(format, G.code + w, core_end, &gd, &d1);
            const int has_prefix = G.code + w != begin;
            if (ok && has_prefix) {
                ok = !scan_generic(format, begin, w, &gd, &d0);
                if (ok && prefix_meets_run(&d0, &S)) ok = 0;
            }
            if (ok && n + 2 <= MAX_CORE_PLANS && G.nplans + 2 <= MAX_CORE_PLANS) {
                const int calc_first = has_prefix && d0.tpdf_calc;
                if (has_prefix) {
                    G.next_tpdf_role = calc_first ? 1 : 0;
                    cp[n] = get_plan_range(format, begin, w);
                    G.next_tpdf_role = 0;
                    if (!cp[n]) { strand_free(&S); return g_err_code; }
                    n++;
                }
                G.next_tpdf_role = calc_first ? 2 : 0;
                G.next_strands = &S.d;
                cp[n] = get_plan_range(format, G.code + w, core_end);
                G.next_strands = 0; G.next_tpdf_role = 0;
                strand_free(&S);
                if (!cp[n]) return g_err_code;
                n++;
                *pn = n;
                return 0;
            }
            strand_free(&S);
        }
        g_err[0] = 0;
    }
    if (G.opt_strand_split && whole->total_chains == 0) ng = split_core(c, starts);
    if (ng > 1 && (G.nplans + ng + 8 > MAX_CORE_PLANS || n + ng + 8 > MAX_CORE_PLANS)) ng = 1;   /* plan table nearly full: whole */
    if (ng > 1) {
        /* Inside a core everything happens frame by frame: a later strand sees what an earlier one stored in
         * the SAME frame (slots, memories, dither state).  Between launches it would see the last frame's.
         * So the pieces of one core must not meet at all; pieces that do are joined again (with whatever
         * lies between them), until the rest are strangers to each other. */
        avdsp_generic_desc gd;
        starts[ng] = dspHeaderPtr->totalLength;
        for (int g = 0; g < ng; g++)
            if (scan_generic(format, G.code + starts[g], starts[g + 1], &gd, &pd[g])) return g_err_code;
        for (int again = 1; again && ng > 1;) {
            again = 0;
            for (int k = 1; k < ng && !again; k++)
                for (int j = 0; j < k; j++)
                    /* the dither value is the one thing a later piece may take from the core's FIRST piece:
                     * the piece with the TPDF_CALC publishes it frame by frame (tpdf_role) */
                    if (cores_meet_ex(&pd[j], &pd[k], j == 0 && pd[0].tpdf_calc && !pd[k].tpdf_calc)) {
                        /* join pieces j..k: drop the starts j+1..k, rescan the joined range */
                        const int gone = k - j;
                        for (int t = j + 1; t + gone <= ng; t++) starts[t] = starts[t + gone];
                        for (int t = j + 1; t + gone < ng; t++) pd[t] = pd[t + gone];
                        ng -= gone;
                        if (scan_generic(format, G.code + starts[j], starts[j + 1], &gd, &pd[j])) return g_err_code;
                        again = 1;
                        break;
                    }
        }
    }
    const int pd_calc_first = ng > 1 && pd[0].tpdf_calc;           /* the first piece holds the TPDF_CALC: it publishes */
    /* Two-way strands: LOAD.., COPYXY, <first way>, SWAPXY, <second way> -- the second way works on the copy of the load's value
     * and never looks at the first way's result, so it is a piece of its own: the load again, then what follows the SWAPXY
     * (two_way_fork; the piece leaves [COPYXY .. SWAPXY] out).  crossoverLV6.bin's second core: its two cascades side by side. */
    typedef struct { int start, end, skip_from, skip_to; } piece_t;
    static piece_t pc[MAX_GROUPS + 2];
    int np = 0;
    const int interpreted = whole->total_chains == 0 && !whole->empty && G.opt_strand_split;
    starts[ng] = dspHeaderPtr->totalLength;
    for (int g = 0; g < ng; g++) {
        const int ps = ng > 1 ? starts[g] : (int)(begin - G.code), pe = ng > 1 ? starts[g + 1] : dspHeaderPtr->totalLength;
        int cw = 0, sw = 0, aw = 0;
        /* (a core stays within MAX_GROUPS pieces, forks included: the per-core entry points take no more) */
        if (interpreted && np + 2 + (ng - g - 1) <= MAX_GROUPS && G.nplans + np + 4 < MAX_CORE_PLANS && n + np + 4 < MAX_CORE_PLANS &&
            two_way_fork(ps, pe, &cw, &sw, &aw)) {
            /* both ways scanned as pieces: they must be strangers like any two pieces of a core */
            static core_deps da, db;
            avdsp_generic_desc gd;
            int ok = !scan_generic(format, G.code + ps, sw, &gd, &da);
            if (ok) {
                G.next_skip_from = cw; G.next_skip_to = aw;
                ok = !scan_generic(format, G.code + ps, pe, &gd, &db);
                G.next_skip_from = G.next_skip_to = 0;
            }
            if (ok && !cores_meet_ex(&da, &db, 1)) {
                pc[np++] = (piece_t){ps, sw, 0, 0};
                pc[np++] = (piece_t){ps, pe, cw, aw};
                continue;
            }
            g_err[0] = 0;
        }
        pc[np++] = (piece_t){ps, pe, 0, 0};
    }
    if (np <= 1) {
        if (n == MAX_CORE_PLANS) return fail(-9, "too many cores");
        if (!whole->empty) cp[n++] = whole;                    /* a shard without chains of this core: nothing to run */
    } else
        for (int g = 0; g < np; g++) {
            if (n == MAX_CORE_PLANS) return fail(-9, "too many cores");
            G.next_tpdf_role = pd_calc_first ? (g == 0 ? 1 : 2) : 0;
            G.next_skip_from = pc[g].skip_from; G.next_skip_to = pc[g].skip_to;
            cp[n] = get_plan_range(format, G.code + pc[g].start, pc[g].end);
            G.next_tpdf_role = 0; G.next_skip_from = G.next_skip_to = 0;
            if (!cp[n]) return g_err_code;
            n++;
        }
    *pn = n;
    return 0;
}

/* (dspRuntimeBlockAllInstancesDevice hands its per-instance strides to block_all through the program's context: inst_call, inst_*_words) */

/* only == 0: every core of the program in program order; else that one core (its strand groups side by side) */
static int block_all(int format, int *rundata, const void *in, int in_stride, int in_io_base,
                     void *out, int out_stride, int out_io_base, int nframes, int on_device, void *stream, int pcm,
                     opcode_t *only)
{
    if (!dspHeaderPtr || !G.code) return fail(-1, "no program loaded");
    if (check_rundata(rundata)) return -1;
    if (nframes <= 0) return 0;
    /* the arrangement is worked out once per lowering (plans live until the next reset / option / parameter upload) */
    arrangement *A = 0;
    for (int i = 0; i < MAX_ARRANGEMENTS; i++)
        if (G.arr[i].valid && G.arr[i].format == format && G.arr[i].only == only) { A = &G.arr[i]; break; }
    if (!A) {
        static core_plan *cp[MAX_CORE_PLANS];
        static int level[MAX_CORE_PLANS];
        int n = 0, nlevels = 0, ncores = 0;
        if (only) {
            ncores = 1;
            if (expand_core(format, only, cp, &n)) return g_err_code;
        } else
            for (int k = 1; k <= MAX_CORE_PLANS; k++) {
                opcode_t *c = dspFindCore(G.code, k);
                if (!c || (k > 1 && c == G.code)) break;
                ncores++;
                if (expand_core(format, c, cp, &n)) return g_err_code;
                if (c == G.code) break;                           /* a program without DSP_CORE is one core */
            }
        if (n == 0 && ncores == 0) return fail(-3, "no cores defined in the program");
        for (int i = 0; i < n; i++) {
            level[i] = 0;
            for (int j = 0; j < i; j++)
                if (level[j] + 1 > level[i] && cores_meet(&cp[j]->deps, &cp[i]->deps)) level[i] = level[j] + 1;
            if (level[i] + 1 > nlevels) nlevels = level[i] + 1;
        }
        if (getenv("AVDSP_DEBUG_LEVELS"))
            for (int i = 0; i < n; i++) {
                fprintf(stderr, "piece %d: words [%d,%d) level %d complete %d calc %d user %d nwr %d nrd %d nranges %d nparams %d meets:", i,
                        (int)(cp[i]->core - G.code), cp[i]->end_word, level[i], cp[i]->deps.complete, cp[i]->deps.tpdf_calc,
                        cp[i]->deps.tpdf_user, cp[i]->deps.nwr, cp[i]->deps.nrd, cp[i]->deps.nranges, cp[i]->deps.nparams);
                for (int j = 0; j < i; j++) if (cores_meet(&cp[j]->deps, &cp[i]->deps)) fprintf(stderr, " %d", j);
                fprintf(stderr, "\n");
            }
        /* slot 0 holds the whole-program arrangement, the others the per-core ones (round robin) */
        int slot = 0;
        if (only) { slot = 1 + G.arr_next; G.arr_next = (G.arr_next + 1) % (MAX_ARRANGEMENTS - 1); }
        A = &G.arr[slot];
        if (n > (only ? MAX_GROUPS : MAX_CORE_PLANS)) return fail(-9, "too many pieces");
        free(A->plans);
        A->plans = (int *)malloc(sizeof(int) * (2 * (size_t)(unsigned)n + 2));
        if (!A->plans) { A->valid = 0; return fail(-9, "out of memory"); }
        A->size = A->plans + n;
        int m = 0;
        for (int l = 0; l < nlevels; l++) {
            A->size[l] = 0;
            for (int i = 0; i < n; i++) if (level[i] == l) { A->plans[m++] = cp[i]->plan_id; A->size[l]++; }
        }
        A->nlevels = nlevels; A->n = n; A->ncores = ncores; A->format = format; A->only = only; A->valid = 1;
    }
    G.last_levels = A->nlevels; G.last_cores = A->ncores; G.last_pieces = A->n;
    G.last_strands = 0;
    for (int i = 0; i < A->n; i++) G.last_strands += avdsp_hip_plan_strands(G.dev, A->plans[i]);
    if (A->n == 0) return 0;                                     /* every core's shard is empty on this rank */
    const int *plans = A->plans, *size = A->size;
    const int nlevels = A->nlevels;
    int rc = G.inst_call
        ? avdsp_hip_run_levels_instances(G.dev, plans, size, nlevels, in, in_stride, in_io_base, G.inst_in_words, out, out_stride, out_io_base,
                                         G.inst_out_words, nframes, stream)
        : on_device
        ? avdsp_hip_run_levels(G.dev, plans, size, nlevels, in, in_stride, in_io_base, out, out_stride, out_io_base,
                               nframes, G.opt_fir_impl, G.opt_biquad_impl, stream)
        : avdsp_hip_run_levels_pcm_host(G.dev, plans, size, nlevels, pcm, in, in_stride, in_io_base, out, out_stride, out_io_base,
                                        nframes, G.opt_fir_impl, G.opt_biquad_impl);
    if (rc) return hip_fail();
    return 0;
}

int dspRuntimeBlockAll(int format, int *rundata, const void *in, int in_stride, int in_io_base,
                       void *out, int out_stride, int out_io_base, int nframes)
{
    (void)ctx_of(rundata); return block_all(format, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes, 0, 0, AVDSP_PCM_S32, 0); }

int dspRuntimeBlockAllDevice(int format, int *rundata, const void *d_in, int in_stride, int in_io_base,
                             void *d_out, int out_stride, int out_io_base, int nframes, void *stream)
{
    (void)ctx_of(rundata); return block_all(format, rundata, d_in, in_stride, in_io_base, d_out, out_stride, out_io_base, nframes, 1, stream, AVDSP_PCM_S32, 0); }

/* ---- instances: N copies of the loaded program side by side (extension) ----
 * The reference runs ONE program per process on one core; a host that has many independent streams for the same program -- a
 * thousand stereo crossovers -- loads it once, asks for N instances and hands over N sample blocks per call.  Instance i starts
 * from a copy of the program's device state (parameters, data area, dither generator, samples[] frame) as it is at the first
 * dspRuntimeBlockAllInstancesDevice call after dspRuntimeSetInstances, and keeps its own from then on; dspRuntimeReset /
 * dspRuntimeUploadState / ...UploadParams afterwards reach instance 0 only -- set the instances again to hand them on.
 * Cores for the frame-parallel interpreter (the reference's programs): the instances are copies of its state; chain cores only: the
 * instances are further chains of the chain kernels' launches; both in one program: every core on the interpreter. */
int dspRuntimeSetInstances(int n)
{
    if (!dspHeaderPtr || !G.code) return fail(-1, "no program loaded");
    if (n < 0 || n > 65536) return fail(-1, "instances: 1 .. 65536 (0: no instances any more)");
    device_current();
    if (G.chain_inst_made) {                             /* chain instances: plans of the old count go, the device keeps instance 0's copy */
        if (replan()) return g_err_code;
        if (G.dev && avdsp_hip_chain_instances(G.dev, 0)) return hip_fail();
        G.chain_inst_made = 0;
    }
    G.inst_chain_mode = 0;
    if (G.dev && avdsp_hip_set_instances(G.dev, n > 0 ? n : 1)) return hip_fail();      /* (else: when the device copy is made) */
    G.ninst = n;
    /* The strand plans know nothing of instances: while a program HAS instances its interpreted cores run as the interpreter's pieces
     * ("strand_lanes" 0, the plans rebuilt once, here -- not silently inside a block call), and dspRuntimeSetInstances(0) gives the
     * program back the arrangement it had. */
    if (n == 0 && G.inst_saved_generic) {                /* (a program of both kinds gets its chain plans back) */
        const int back = G.inst_saved_generic - 1;
        G.inst_saved_generic = 0;
        if (set_option_here("generic", back)) return g_err_code;
    }
    if (n >= 1 && G.opt_strand_lanes != 0) {
        G.inst_saved_lanes = 1 + G.opt_strand_lanes;
        return set_option_here("strand_lanes", 0);
    }
    if (n == 0 && G.inst_saved_lanes) {
        const int back = G.inst_saved_lanes - 1;
        G.inst_saved_lanes = 0;
        return set_option_here("strand_lanes", back);
    }
    return 0;
}

int dspRuntimeBlockAllInstancesDevice(int format, int *rundata, const void *d_in, int in_stride, int in_io_base, size_t in_inst_words,
                                      void *d_out, int out_stride, int out_io_base, size_t out_inst_words, int nframes, void *stream)
{
    (void)ctx_of(rundata);
    if (G.ninst < 1) return fail(-1, "dspRuntimeSetInstances first");
    if (!dspHeaderPtr || !G.code) return fail(-1, "no program loaded");
    if (format < 2 || format > 6) return fail(-1, "DSP_FORMAT %d is not one of 2..6", format);
    /* Which kind of program?  Every core a set of independent chains (lower_core takes it): the instances are further chains of the
     * chain kernels' launches.  Every core one for the interpreter: the instances are further copies of its state (round 4).  A
     * program with both is refused -- its instances would keep their state in two places. */
    if (!G.inst_chain_mode) {
        int nchain = 0, nother = 0;
        if (!G.have_rate) return fail(-1, "dspRuntimeReset(fs) has not selected a sample rate yet");
        for (int k = 1; k <= MAX_CORE_PLANS; k++) {
            opcode_t *c = dspFindCore(G.code, k);
            if (!c || (k > 1 && c == G.code)) break;
            lowered L;
            memset(&L, 0, sizeof L);
            const int rc = G.opt_generic ? -8 : lower_core(format, dspFindCoreBegin(c), &L);
            lowered_free(&L);
            if (rc == 0) nchain++; else if (rc == -8) nother++; else return g_err_code;
            if (c == G.code) break;
        }
        if (nchain && nother) {
            /* both kinds: the instances' state must live in ONE place, and the interpreter runs chain cores too -- every core on it for as
             * long as the program has instances ("generic" 1: the chain plans go, their FIR histories home into the mirror first;
             * dspRuntimeSetInstances(0) gives the program its arrangement back) */
            G.inst_saved_generic = 1 + G.opt_generic;
            if (set_option_here("generic", 1)) return g_err_code;
        }
        G.inst_chain_mode = nchain && !nother ? 2 : 1;
    }
    if (G.inst_chain_mode == 2) {
        if (check_rundata(rundata)) return -1;
        if (nframes <= 0) return 0;
        if (G.shard_world > 1) return fail(-1, "instances of a sharded program: shard the instances instead");
        if (in_stride <= 0 || out_stride <= 0) return fail(-1, "instances: strides in words per frame, please");
        /* the kernels' sample offsets are 32 bits: the last instance's block must end below 4 GiB from the first one's start */
        if (((unsigned long long)(G.ninst - 1) * in_inst_words + (unsigned long long)nframes * (unsigned)in_stride) >= (1ull << 30) ||
            ((unsigned long long)(G.ninst - 1) * out_inst_words + (unsigned long long)nframes * (unsigned)out_stride) >= (1ull << 30))
            return fail(-1, "instances: the sample blocks of %d instances span more than 2^30 words", G.ninst);
        if (G.ninst > 1 && (in_inst_words < (size_t)nframes * (size_t)in_stride || out_inst_words < (size_t)nframes * (size_t)out_stride))
            return fail(-1, "instances: a block of %d frames does not fit the distance between two instances' blocks", nframes);
        if (G.ninst > 1 && (!G.chain_inst_made || G.chain_inst_in != in_inst_words || G.chain_inst_out != out_inst_words)) {
            /* (first call, or other block distances: the plans are made for them) */
            if (replan()) return g_err_code;              /* (the instances' FIR histories go home into their mirror copies) */
            if (ensure_device()) return g_err_code;
            /* the copies are made ONCE, from what instance 0's state is then; other block distances only re-make the plans */
            if (!G.chain_inst_made && avdsp_hip_chain_instances(G.dev, G.ninst)) return hip_fail();
            G.chain_inst_made = 1; G.chain_inst_in = in_inst_words; G.chain_inst_out = out_inst_words;
        }
        int rc = 0;
        G.inst_call = 1;
        for (int k = 1; k <= MAX_CORE_PLANS && !rc; k++) {
            opcode_t *c = dspFindCore(G.code, k);
            if (!c || (k > 1 && c == G.code)) break;
            core_plan *cp = get_plan(format, c);
            if (!cp) { rc = g_err_code; break; }
            const int same_win = G.chain_inst_win[0] == format && G.chain_inst_win[1] == in_io_base && G.chain_inst_win[2] == in_stride &&
                                 G.chain_inst_win[3] == out_io_base && G.chain_inst_win[4] == out_stride;
            if (!cp->empty && same_win) {                 /* (checked by an earlier call: the usual case, a host's windows do not move) */
                if (avdsp_hip_run_block(G.dev, cp->plan_id, d_in, in_stride, in_io_base, d_out, out_stride, out_io_base,
                                        nframes, G.opt_fir_impl, G.opt_biquad_impl, stream)) rc = hip_fail();
            } else if (!cp->empty) {
                /* the windows against the program's own IO numbers (the plan's carry the instances' offsets) */
                lowered L;
                memset(&L, 0, sizeof L);
                if (lower_core(format, dspFindCoreBegin(c), &L)) { lowered_free(&L); rc = g_err_code; break; }
                for (int i = 0; i < L.nchains && !rc; i++) {
                    if (L.chains[i].in_io < in_io_base || L.chains[i].in_io >= in_io_base + in_stride) rc = fail(-10, "input window IO [%d,%d) does not cover IO %d", in_io_base, in_io_base + in_stride, L.chains[i].in_io);
                    for (int o = 0; o < L.chains[i].n_out && !rc; o++)
                        if (L.chains[i].out_io[o] < out_io_base || L.chains[i].out_io[o] >= out_io_base + out_stride) rc = fail(-10, "output window IO [%d,%d) does not cover IO %d", out_io_base, out_io_base + out_stride, L.chains[i].out_io[o]);
                }
                lowered_free(&L);
                if (!rc && avdsp_hip_run_block(G.dev, cp->plan_id, d_in, in_stride, in_io_base, d_out, out_stride, out_io_base,
                                               nframes, G.opt_fir_impl, G.opt_biquad_impl, stream)) rc = hip_fail();
            }
            if (c == G.code) break;
        }
        G.inst_call = 0;
        if (!rc) { G.chain_inst_win[0] = format; G.chain_inst_win[1] = in_io_base; G.chain_inst_win[2] = in_stride; G.chain_inst_win[3] = out_io_base; G.chain_inst_win[4] = out_stride; }
        else G.chain_inst_win[0] = 0;
        return rc;
    }
    /* the strand plans know nothing of instances: dspRuntimeSetInstances(n > 1) has switched them off for as long as the program has
     * instances (a caller who turned them on again since gets told, not overridden) */
    if (G.opt_strand_lanes != 0)
        return fail(-1, "\"strand_lanes\" was set again after dspRuntimeSetInstances(%d): instances run on the interpreter's pieces", G.ninst);
    G.inst_call = 1; G.inst_in_words = in_inst_words; G.inst_out_words = out_inst_words;
    const int rc = block_all(format, rundata, d_in, in_stride, in_io_base, d_out, out_stride, out_io_base, nframes, 1, stream, AVDSP_PCM_S32, 0);
    G.inst_call = 0;
    return rc;
}

/* the data area of instance i (dataSize words), as dspRuntimeSyncState brings back instance 0's into the caller's buffer */
int dspRuntimeInstanceState(int inst, int *dst)
{
    if (!dspHeaderPtr || !G.dev) return fail(-1, "no program loaded, or no block has run yet");
    device_current();
    if (avdsp_hip_download_instance_words(G.dev, inst, dst, (int)dspHeaderPtr->totalLength, (int)dspHeaderPtr->dataSize)) return hip_fail();
    return 0;
}

/* linux/avdsp_plugin.c:95-142 whole: packed PCM in (:109-121), every core, S32 out */
int dspRuntimeBlockAllPcm(int format, int *rundata, int pcm, const void *src, int in_stride, int in_io_base,
                          int *dst, int out_stride, int out_io_base, int nframes)
{
    (void)ctx_of(rundata);
    if (format != 2 && format != 3 && format != 4)
        return fail(-1, "packed PCM feeds the int-sample formats 2, 3 and 4 (DSP_FORMAT %d has float samples)", format);
    return block_all(format, rundata, src, in_stride, in_io_base, dst, out_stride, out_io_base, nframes, 0, 0, pcm, 0);
}

/* linux/avdsp_plugin.c:95-142 with the sample-format switch of :109-121: packed PCM in, S32 out */
int dspRuntimeBlockPcm(int format, opcode_t *core, int *rundata, int pcm, const void *src, int in_stride, int in_io_base,
                       int *dst, int out_stride, int out_io_base, int nframes)
{
    (void)ctx_of(core);
    if (format != 2 && format != 3 && format != 4)
        return fail(-1, "packed PCM feeds the int-sample formats 2, 3 and 4 (DSP_FORMAT %d has float samples)", format);
    core_plan *cp = get_plan(format, core);
    if (!cp) return g_err_code;
    if (check_rundata(rundata)) return -1;
    if (nframes <= 0 || cp->empty) return 0;
    if (takes_pieces(cp, nframes, in_stride, in_io_base, out_stride, out_io_base))
        return block_all(format, rundata, src, in_stride, in_io_base, dst, out_stride, out_io_base, nframes, 0, 0, pcm, core);
    if (avdsp_hip_run_block_pcm_host(G.dev, cp->plan_id, pcm, src, in_stride, in_io_base, dst, out_stride, out_io_base,
                                     nframes, G.opt_fir_impl, G.opt_biquad_impl))
        return hip_fail();
    return 0;
}

/* ---- linux/avdsp_plugin.c:133-137 ---- */
int dspRuntimeTagOutputDevice(void *d_out, int out_stride, int column, int nframes, void *stream)
{
    if (!G.dev) return fail(-1, "no device program yet: run a block first");
    device_current();
    if (column < 0 || column >= out_stride) return fail(-1, "tag column %d outside the output window of %d", column, out_stride);
    if (avdsp_hip_tag_output(G.dev, (int *)d_out + column, out_stride, nframes, 0, 0, stream)) return hip_fail();
    return 0;
}

int dspRuntimeTagOutputReset(int previoussample)
{
    if (!G.dev) return fail(-1, "no device program yet: run a block first");
    device_current();
    if (avdsp_hip_tag_output(G.dev, 0, 0, 0, 1, previoussample, 0)) return hip_fail();
    return 0;
}

int dspRuntimeTagOutput(int *out, int out_stride, int column, int nframes)
{
    if (!G.dev) return fail(-1, "no device program yet: run a block first");
    device_current();
    if (column < 0 || column >= out_stride) return fail(-1, "tag column %d outside the output window of %d", column, out_stride);
    if (nframes <= 0) return 0;
    /* one column through the device: the carried value lives there */
    int *col = (int *)malloc((size_t)nframes * sizeof(int));
    if (!col) return fail(-9, "out of memory");
    for (int n = 0; n < nframes; n++) col[n] = out[(size_t)n * out_stride + column];
    int rc = avdsp_hip_tag_column_host(G.dev, col, nframes);
    if (!rc) for (int n = 0; n < nframes; n++) out[(size_t)n * out_stride + column] = col[n];
    free(col);
    if (rc) return hip_fail();
    return 0;
}

int dspRuntimeUnpackPcmDevice(int pcm, const void *d_src, int *d_dst, long long nsamples, void *stream)
{
    if (!G.dev) return fail(-1, "no device program yet: run a block first");
    device_current();
    if (avdsp_hip_unpack_pcm(G.dev, pcm, d_src, d_dst, (size_t)nsamples, stream)) return hip_fail();
    return 0;
}

/* One frame = a block of one frame whose input and output windows are both the caller's samples[]
 * array (IO numbers index it directly).  The window is the span of IO numbers the core touches. */
static int one_frame(int format, opcode_t *core, int *rundata, void *samples)
{
    (void)ctx_of(core);
    core_plan *cp = get_plan(format, core);
    if (!cp) return g_err_code;
    /* the device side knows the IO span of the plan: stride 0 asks it to use that span */
    return block_host(format, core, rundata, samples, 0, 0, samples, 0, 0, 1);
}

int dspRuntime_2(opcode_t *core, int *rundata, int *samples)   { return one_frame(2, core, rundata, samples); }
int dspRuntime_3(opcode_t *core, int *rundata, int *samples)   { return one_frame(3, core, rundata, samples); }
int dspRuntime_4(opcode_t *core, int *rundata, int *samples)   { return one_frame(4, core, rundata, samples); }
int dspRuntime_5(opcode_t *core, int *rundata, float *samples) { return one_frame(5, core, rundata, samples); }
int dspRuntime_6(opcode_t *core, int *rundata, float *samples) { return one_frame(6, core, rundata, samples); }

int dspRuntimeSyncState(int *rundata)
{
    (void)ctx_of(rundata);
    if (!dspHeaderPtr) return fail(-1, "no program loaded");
    if (check_rundata(rundata)) return -1;
    if (!G.dev || !G.dev_state_valid) return 0;              /* nothing ran yet: host copy is current */
    /* the header stays the host's: only words behind it can have been written (DSP_STORE_MEM) */
    const int first = (int)(sizeof(dspHeader_t) / sizeof(int));
    if (avdsp_hip_download_words(G.dev, (int32_t *)G.code, first, G.total_words - first))
        return hip_fail();
    return 0;
}

/* Live parameter control: the reference reads gains, coefficients, delays ... from the program words on
 * every frame, so a host may edit them in place between frames.  Here the words live in the device mirror
 * and some are folded into the plans; this call carries the host's edits over without touching the state. */
int dspRuntimeUploadParams(void)
{
    if (!dspHeaderPtr) return fail(-1, "no program loaded");
    if (!G.dev) return 0;                                     /* nothing on the device yet: the next block uploads everything */
    device_current();
    const int first = (int)(sizeof(dspHeader_t) / sizeof(int));
    if (avdsp_hip_prog_clear_plans(G.dev) ||
        avdsp_hip_upload_words(G.dev, (const int32_t *)G.code, first, dspHeaderPtr->totalLength - first))
        return hip_fail();
    G.nplans = 0;                                             /* cores are lowered again at their next block */
    for (int i_ = 0; i_ < MAX_ARRANGEMENTS; i_++) G.arr[i_].valid = 0;
    return 0;
}

int dspRuntimeUploadState(const int *rundata)
{
    (void)ctx_of(rundata);
    if (!dspHeaderPtr) return fail(-1, "no program loaded");
    if (check_rundata(rundata)) return -1;
    if (!G.dev) return 0;                                     /* uploaded with the whole buffer at first use */
    if (avdsp_hip_upload_words(G.dev, (const int32_t *)G.code, dspHeaderPtr->totalLength, dspHeaderPtr->dataSize))
        return hip_fail();
    return 0;
}
