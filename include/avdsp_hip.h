/*
 * avdsp_hip.h -- the thin C ABI between the C host runtime (avdsp_amd/csrc/avdsp_host.c) and the
 * hand-written gfx950 kernels (avdsp_amd/csrc/avdsp_kernels.hip).  Plain pointers and sizes only.
 *
 * The host side lowers one DSP core (dsp_runtime.c:302-1314 restricted to channel-independent
 * chains) into an avdsp_plan_desc; the device side keeps a mirror of the caller's buffer
 * (program words + state area, same word layout as on the host) and runs
 *
 *     LOAD | LOAD_GAIN  ->  BIQUADS cascade  ->  [FIR]  ->  [SAT0DB]  ->  STORE
 *     (dsp_runtime.c:565-607, 827-849 + dsp_biquadSTD.h:25-119, 928-969 + dsp_firSTD.h:38-52,
 *      464-475, 610-633)
 *
 * for every chain of the core over a block of frame-interleaved samples.
 */
#ifndef AVDSP_HIP_H_
#define AVDSP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVDSP_MAX_STORES 4

/* how the accumulator is loaded at the head of a chain */
enum { AVDSP_LOAD_PLAIN = 0, AVDSP_LOAD_GAIN = 1 };

/* One channel chain.  All *_word fields are absolute word indices into the caller's buffer
 * (program words first, state area from word `totalLength`).                                    */
typedef struct avdsp_chain {
    int32_t  in_io;                  /* IO number read by LOAD / LOAD_GAIN                       */
    int32_t  load_mode;              /* AVDSP_LOAD_*                                             */
    uint32_t gain_bits;              /* Q4.28 int or float bits of the LOAD_GAIN parameter       */
    int32_t  nsec;                   /* biquad sections in cascade order (banks concatenated)    */
    int32_t  sec_base;               /* first entry of this chain in sec_coef_word/sec_state_word */
    int32_t  fir_taps;               /* 0 = no FIR                                               */
    int32_t  fir_coef_word;          /* first tap (the word after the length word)               */
    int32_t  fir_state_word;         /* FIR delay line, fir_taps words, st[i] = x[n-1-i]         */
    int32_t  sat;                    /* SAT0DB in front of the stores                            */
    int32_t  n_out;                  /* number of STOREs (same value to each)                    */
    int32_t  out_io[AVDSP_MAX_STORES];
} avdsp_chain;

/* words between two copies of the mirror (chain instances, avdsp_hip_chain_instances): even, so that state words keep their alignment */
#define AVDSP_INSTANCE_STRIDE(total_words) (((total_words) + 1) & ~1)
typedef struct avdsp_plan_desc {
    int32_t  format;                 /* 2, 4 or 6                                                */
    int32_t  nchains;
    const avdsp_chain *chains;
    int32_t  nsections;              /* total over chains                                        */
    const int32_t *sec_coef_word;    /* per section: word index of b0 for the CURRENT sample rate */
    const int32_t *sec_state_word;   /* per section: word index of its 6 state words             */
    int32_t  store_mask;             /* tpdf mask applied by STORE in int-sample formats         */
    int32_t  instances;              /* > 1: the chains are that many copies of a core's, copy i addressing mirror copy i (word indices
                                        + i * total words) and sample block i (IO numbers + i * the blocks' distance in words):
                                        avdsp_hip_chain_instances() must have made the copies; 0 / 1: an ordinary plan */
} avdsp_plan_desc;

/* A core that is not a set of independent chains runs through the general device interpreter
 * (avdsp_interp.inc: every opcode of dsp_runtime.c:302-1314, formats 2..6, sequential over frames).
 * The host has validated every offset the opcode stream can reach before it asks for this.       */
typedef struct avdsp_generic_desc {
    int32_t  format;                 /* 2..6                                                     */
    int32_t  core_word;              /* word index of the first opcode executed                  */
    int32_t  end_word;               /* execution stops in front of this opcode word (a core cut into groups of
                                        strands that do not depend on each other); 0 = up to DSP_CORE / END    */
    int32_t  prog_words;             /* header totalLength: the data area starts at this word    */
    int32_t  freq_index, num_freq;   /* dsp_runtime.c:103-107                                    */
    int32_t  biquad_freq_skip;       /* 2 + 6*num_freq, :37                                      */
    int32_t  biquad_freq_offset;     /* 5 + 6*freq_index, :110                                   */
    uint32_t delay_line_factor;      /* 2^32 * fs / 10^6, :82-90,108                             */
    int32_t  io_span;                /* highest IO number the core touches + 1                   */
    int32_t  io_in_min, io_in_max, io_out_min, io_out_max;   /* for the window check (max < min = none) */
    /* Frame-parallel execution (avdsp_interp.inc, interp_wave: 64 frames of a block side by side, one per
     * lane).  wave_ok = the host found nothing that one frame hands to the next outside opcode-private
     * state; the remaining condition is checked per block call, against the caller's windows:           */
    int32_t  wave_ok;
    uint32_t carried_io[8];          /* bit io: the core reads samples[io] before it stores it; such a slot must lie
                                        in the block's input or output window (else it is last frame's value)   */
    int32_t  nvm, vm_word[16];       /* mirror words written and read back inside one frame (STORE_MEM -> LOAD_MEM,
                                        LOAD_MUX / TPDF result -> LOAD_MEM_DATA): kept per lane                */
    int32_t  seq_words;              /* longest DSP_FIR impulse + 64 (0 = no FIR)                              */
    /* What the core owns: an interpreter launch writes back only this, so that cores which do not meet may run
     * side by side (avdsp_hip_run_level).  nown < 0: unknown, the launch writes everything back and runs alone. */
    int32_t        nown;             /* pairs in own[]                                                          */
    const int32_t *own;              /* mirror word ranges [lo, hi): opcode state, STORE_MEM / result words     */
    uint32_t early_io[8];            /* frame slots < 256 the core reads before it has stored them in the frame  */
    uint32_t written_io[8];          /* frame slots < 256 the core stores                                        */
    int32_t  tpdf_calc;              /* the core holds the DSP_TPDF_CALC: it alone writes the dither globals     */
    int32_t  tpdf_role;              /* pieces of a core cut into strand groups: 1 = the piece with the TPDF_CALC leaves
                                        every frame's dither value for the later pieces (2) of that core; 0 otherwise    */
    int32_t  skip_from, skip_to;     /* a piece that leaves a stretch of its range out: execution jumps from opcode word skip_from
                                        to skip_to (the second way of a two-way strand: its load, then what follows the SWAPXY);
                                        0 = nothing left out                                                                   */
    int32_t  dither_only;            /* the stretch is a DSP_TPDF_CALC and nothing else (the piece in front of a strand run):  */
    int32_t  dither_arg, dither_result_word;   /* its dither width word and the mirror word its result goes to (tpdf_walk)   */
} avdsp_generic_desc;

/* A loaded program on the device: the mirror of the caller's buffer plus one plan per lowered core */
typedef struct avdsp_hip_prog avdsp_hip_prog;

/* all functions: 0 / non-NULL / id >= 0 on success; on failure a negative code / NULL and a message
 * in avdsp_hip_last_error()                                                                     */
int             avdsp_hip_device_count(void);
int             avdsp_hip_set_device(int ordinal);
avdsp_hip_prog *avdsp_hip_prog_create(int total_words);
void            avdsp_hip_prog_destroy(avdsp_hip_prog *prog);
int             avdsp_hip_prog_add_plan(avdsp_hip_prog *prog, const avdsp_plan_desc *desc);   /* plan id */
int             avdsp_hip_prog_add_generic(avdsp_hip_prog *prog, const avdsp_generic_desc *desc);   /* plan id */
/* dspTpdfInit (dsp_tpdf.h:85-99) on the device copy of the dither generator's globals */
int             avdsp_hip_tpdf_reset(avdsp_hip_prog *prog, int seed, int default_dither);

/* forget every plan but keep the mirror (the FIR histories are written back into it first): the host
 * has edited parameters in the program words and will lower its cores again */
int             avdsp_hip_prog_clear_plans(avdsp_hip_prog *prog);

/* mirror maintenance: word ranges of the caller's buffer */
int avdsp_hip_upload_words(avdsp_hip_prog *prog, const int32_t *host_buf, int first_word, int nwords);
int avdsp_hip_download_words(avdsp_hip_prog *prog, int32_t *host_buf, int first_word, int nwords);
int avdsp_hip_zero_words(avdsp_hip_prog *prog, int first_word, int nwords);

/* one block; d_in/d_out are device pointers to frame-interleaved 32-bit samples; asynchronous on
 * `stream` (hipStream_t as void*).  fir_impl / biquad_impl: see dspRuntimeSetOption.            */
int avdsp_hip_run_block(avdsp_hip_prog *prog, int plan, const void *d_in, int in_stride, int in_io_base,
                        void *d_out, int out_stride, int out_io_base, int nframes,
                        int fir_impl, int biquad_impl, void *stream);

/* Several cores over one block: plans[] in program order, grouped into levels; the cores of a level do not meet
 * (host's analysis) and run at the same time on side streams, the levels one after the other.  Same meaning as
 * calling avdsp_hip_run_block for every plan in order.                                           */
int avdsp_hip_run_levels(avdsp_hip_prog *prog, const int *plans, const int *level_size, int nlevels,
                         const void *d_in, int in_stride, int in_io_base, void *d_out, int out_stride, int out_io_base,
                         int nframes, int fir_impl, int biquad_impl, void *stream);
int avdsp_hip_run_levels_host(avdsp_hip_prog *prog, const int *plans, const int *level_size, int nlevels,
                              const void *h_in, int in_stride, int in_io_base, void *h_out, int out_stride, int out_io_base,
                              int nframes, int fir_impl, int biquad_impl);

int avdsp_hip_run_levels_pcm_host(avdsp_hip_prog *prog, const int *plans, const int *level_size, int nlevels, int pcm,
                                  const void *h_src, int in_stride, int in_io_base, void *h_out, int out_stride, int out_io_base,
                                  int nframes, int fir_impl, int biquad_impl);     /* packed PCM in (AVDSP_PCM_*), unpacked once */

/* host-buffer convenience: stages in/out through device scratch and synchronises               */
int avdsp_hip_run_block_host(avdsp_hip_prog *prog, int plan, const void *h_in, int in_stride, int in_io_base,
                             void *h_out, int out_stride, int out_io_base, int nframes,
                             int fir_impl, int biquad_impl);

/* The same as a queue (avdsp_kernels.hip: up to 4 blocks in flight, copies of one block under the kernels of another).
 * submit returns the number of blocks in flight (0: done on the spot -- interpreter core or a block under 256 frames),
 * -1 on error; the caller's buffers are pinned in place and must stay allocated and untouched until wait has let the
 * block through.  wait blocks until at most max_in_flight submitted blocks are unfinished (oldest first) and returns
 * how many still are.                                                                                            */
int avdsp_hip_submit_block_host(avdsp_hip_prog *prog, int plan, const void *h_in, int in_stride, int in_io_base,
                                void *h_out, int out_stride, int out_io_base, int nframes, int fir_impl, int biquad_impl);
int avdsp_hip_wait_block_host(avdsp_hip_prog *prog, int max_in_flight);

/* Host sample formats of linux/avdsp_plugin.c:103-121 in front of a block: packed little-endian PCM,
 * frame-interleaved, unpacked on the device to the 32-bit s.31 words the int-sample formats (2, 3, 4) load:
 * S32 as is, S24_3LE bytes b0 b1 b2 -> b0<<8 | b1<<16 | b2<<24, S16 -> sample << 16.                    */
enum { AVDSP_PCM_S32 = 0, AVDSP_PCM_S24_3LE = 1, AVDSP_PCM_S16 = 2 };
int avdsp_hip_unpack_pcm(avdsp_hip_prog *prog, int pcm, const void *d_src, void *d_dst, size_t nsamples, void *stream);
/* host buffers: src packed PCM [nframes][in_stride], dst int32 [nframes][out_stride] */
int avdsp_hip_run_block_pcm_host(avdsp_hip_prog *prog, int plan, int pcm, const void *h_src, int in_stride, int in_io_base,
                                 void *h_out, int out_stride, int out_io_base, int nframes,
                                 int fir_impl, int biquad_impl);

/* Per-kernel timing: when enabled, every kernel launch of run_block is bracketed by a HIP event pair
 * recorded on the launch stream; profile_read waits for the recorded pairs of one kind, returns
 * the summed duration and the number of launches, and forgets them.  on: 0 = off, 1 = every kind,
 * 2 * mask = only the kinds whose bit is set in mask (an event pair costs a few microseconds of stream
 * time: a benchmark times the kernel it reports and nothing else).                               */
enum { AVDSP_KERNEL_BIQUAD = 0, AVDSP_KERNEL_FIR = 1, AVDSP_KERNEL_PASS = 2, AVDSP_KERNEL_GENERIC = 3, AVDSP_KERNEL_UNPACK = 4, AVDSP_KERNEL_GENERIC_WAVE = 5, AVDSP_KERNEL_STRANDS = 6 };
/* N instances of the loaded program side by side (dspRuntimeSetInstances / dspRuntimeBlockAllInstancesDevice): instance i has a copy of
 * the whole device state of its own and its sample blocks at d_in + i * in_inst_words, d_out + i * out_inst_words (32-bit words). */
int avdsp_hip_set_instances(avdsp_hip_prog *prog, int n);
int avdsp_hip_run_levels_instances(avdsp_hip_prog *prog, const int *plans, const int *level_size, int nlevels,
                                   const void *d_in, int in_stride, int in_io_base, size_t in_inst_words,
                                   void *d_out, int out_stride, int out_io_base, size_t out_inst_words, int nframes, void *stream);
int avdsp_hip_download_instance_words(avdsp_hip_prog *p, int inst, int32_t *host_buf, int first, int n);
int avdsp_hip_profile_enable(avdsp_hip_prog *prog, int on);
int avdsp_hip_profile_read(avdsp_hip_prog *prog, int kind, double *total_ms, int *launches);
/* of the launches the latest avdsp_hip_profile_read of `kind` summed: how many were timed by an event pair recorded around the launch
 * (which reads a few microseconds more than the kernel took) instead of the dispatch's own start / end stamps */
int avdsp_hip_profile_last_pairs(avdsp_hip_prog *prog, int kind);

/* Strand plans (round 3): a stretch of an interpreted core that is N repetitions of ONE opcode sequence with different operands
 * -- one strand per channel: LOAD[_GAIN] .. GAIN / DELAY / BIQUADS / X-Y moves .. SAT0DB[_TPDF][_GAIN] .. STORE, the shape of the
 * reference's crossover programs -- runs that sequence once per frame batch with LANE = STRAND instead of one wave per strand group.
 * The host proves the strands identical in shape and strangers to each other (avdsp_host.c strand_lower) and hands over the sequence
 * as micro-operations plus one row of arguments per strand; attached to the stretch's generic plan, which stays the fallback for
 * calls whose windows the strand kernel does not take (windows that share IO numbers, a STORE into the input window).           */
enum { AVDSP_SOP_LOAD = 1, AVDSP_SOP_LOAD_GAIN, AVDSP_SOP_GAIN, AVDSP_SOP_COPYXY, AVDSP_SOP_SWAPXY, AVDSP_SOP_COPYYX, AVDSP_SOP_ADDXY, AVDSP_SOP_ADDYX,
       AVDSP_SOP_SUBXY, AVDSP_SOP_SUBYX, AVDSP_SOP_NEGX, AVDSP_SOP_SHIFT, AVDSP_SOP_SAT0DB, AVDSP_SOP_SAT0DB_TPDF, AVDSP_SOP_SAT0DB_GAIN,
       AVDSP_SOP_SAT0DB_TPDF_GAIN, AVDSP_SOP_STORE, AVDSP_SOP_LOAD_MEM, AVDSP_SOP_STORE_MEM, AVDSP_SOP_DELAY, AVDSP_SOP_DELAY_DP, AVDSP_SOP_BIQUADS };
/* a0..a2: columns of the strand's argument row (values there: IO numbers, absolute word indices into the mirror, payload values);
 * imm: the one operand the shape makes uniform (BIQUADS: sections per bank); rcol: first of the operation's columns in the
 * kernel's per-lane table of resolved operands and running state (LDS; filled once per launch, avdsp_strand_rcols() of them) */
typedef struct avdsp_strand_op { int32_t op, a0, a1, a2, imm, rcol; } avdsp_strand_op;
typedef struct avdsp_strand_desc {
    int32_t nops; const avdsp_strand_op *ops;
    int32_t nstrands, nargs; const int32_t *args;          /* [nstrands][nargs] */
    int32_t stored_io_max;                                 /* highest IO a strand stores (the scratch frame must reach it) */
    int32_t nres;                                          /* columns of the resolved table: sum of avdsp_strand_rcols over the operations */
} avdsp_strand_desc;
/* resolved columns of one operation (format: 64-bit accumulator models keep memory values in two words) */
static inline int avdsp_strand_rcols(int op, int imm, int alu_words)
{
    switch (op) {
    case AVDSP_SOP_LOAD: case AVDSP_SOP_STORE: case AVDSP_SOP_GAIN: case AVDSP_SOP_SHIFT: case AVDSP_SOP_SAT0DB_GAIN:
    case AVDSP_SOP_SAT0DB_TPDF_GAIN: case AVDSP_SOP_STORE_MEM: return 1;
    case AVDSP_SOP_LOAD_GAIN: return 2;
    case AVDSP_SOP_LOAD_MEM: return alu_words;
    case AVDSP_SOP_DELAY: case AVDSP_SOP_DELAY_DP: return 4;           /* n, allocation, word of the line's counter, running position */
    case AVDSP_SOP_BIQUADS: return 2 + 11 * imm;                        /* bypass, state word, per section 5 coefficients + 6 state words */
    }
    return 0;
}
int avdsp_hip_plan_add_strands(avdsp_hip_prog *prog, int plan, const avdsp_strand_desc *d);
int avdsp_hip_plan_strands(const avdsp_hip_prog *prog, int plan);      /* strands of the plan (0 = none) */

/* Launch arrangement of the chain kernels.  AVDSP_OPT_OVERLAP 1: the cascade of block k+1 may run under the FIR of
 * block k (side stream; see launch_all in avdsp_kernels.hip) -- the caller then guarantees that a block's input is
 * complete in memory when the call is made.                                                                     */
enum { AVDSP_OPT_OVERLAP = 0, AVDSP_OPT_PROFILE_STRIDE = 1, AVDSP_OPT_FIR_ROWS = 3, AVDSP_OPT_HOST_SPLIT = 4, AVDSP_OPT_HOST_PIN = 5,
       AVDSP_OPT_READY_WORDS = 6, AVDSP_OPT_LANE_HW = 7, AVDSP_OPT_FIR_SPLIT = 8, AVDSP_OPT_FIR_LAUNCH = 9, AVDSP_OPT_FIR_LEAN = 10, AVDSP_OPT_RING_WAIT = 11,
       AVDSP_OPT_SIDE_BY_SIDE = 13, /* read-only: 1 kernels of two queues were seen to run at once (the precondition of ready words), 0 not (dispatches are serialised: a PMC profiler, a debugger), -1 not probed yet */
       AVDSP_OPT_GROUP_FANOUT = 17, /* 1 (default): a plan's cascade launches -- one per section count -- go out over up to four streams side by side; 0: one after the other */
       AVDSP_OPT_CU_SPLIT = 16,     /* experiment: the overlap mode's cascades on that many CUs of their own (CU-masked stream), the FIRs on the others */
       AVDSP_OPT_STREAMS_REMADE = 15, /* read-only: how many times the cascades' stream was made anew because it shared a hardware queue with the FIRs' */
       AVDSP_OPT_READY_MODE = 14,   /* read-only: how the latest overlapped launch's FIR found its cascades' block: 0 event, 1 / 2 ready words */
       AVDSP_OPT_READY_TEST = 12 /* tests only: that many coming "ready_words" 2 launches never get their words set (their FIR waves time out) */ };
/* FIR_LEAN: fir_tile's chunk boundary with a third of the vector instructions: -1 by the plan (default), 0 never, 1 always. */
/* READY_WORDS (under OVERLAP): how a block's FIR finds its cascades' block in the rings: 0 an event between the two queues, 1 per-chain
 * words published by the cascade's waves (write-through stores) and polled by the FIR's, 2 the words set by a kernel behind the
 * cascade on the cascades' stream (no wait packet on the FIRs' stream, no acquire in the FIR), -1 (default) 2 where the FIR is the
 * bound (a launch of more than one round of waves), else 0. */
/* RING_WAIT (under OVERLAP): who waits for the FIR three blocks back before a block's cascade may append to the rings: 1 (default) the
 * host, polling that FIR's event for at most 1 ms before it enqueues the cascade (then the stream after all), 0 the cascades' stream. */
/* FIR_LAUNCH: how the overlap mode enqueues a FIR launch (avdsp_hip_prog::fir_launch_mode): -1 auto (default), 0, 1, 2. */
/* FIR_SPLIT 1 (opt-in; default 0): a fir_tile launch that leaves a SIMD one wave at most (256 chains x 4096 taps) cuts every tile's taps
 * over two waves and adds the two partial sums -- within BASELINE's 1e-6 of the reference, not its bits any more. */
/* LANE_HW 1 (default): formats 3 and 5 multiply with v_mul_f32 under round-toward-zero wherever the operands' exponents make that the
 * reference's dspMulFloatFloat bit for bit (fir_lane_hw, chain_rows); 0: the integer restatement of the product throughout. */
/* READY_WORDS 1: under OVERLAP the FIR finds its cascades' blocks through per-chain ready words polled inside the kernel instead of
 * an event between the two queues (0, the default: the event -- the words measured slower on every configuration). */
int avdsp_hip_chain_instances(avdsp_hip_prog *prog, int n);   /* the mirror n times side by side (no plan may exist); <= 1: one copy again */
int avdsp_hip_prog_get_option(avdsp_hip_prog *prog, int key);   /* AVDSP_OPT_SIDE_BY_SIDE, AVDSP_OPT_READY_MODE */
int avdsp_hip_ready_clear(avdsp_hip_prog *prog);       /* the caller acknowledges the time-outs: count and sticky mark start again */
int avdsp_hip_last_error_is_ready_timeout(void);       /* 1: the latest failure of this thread was the sticky ready-word time-out (the host maps it to -11) */
int avdsp_hip_ready_timeouts(avdsp_hip_prog *prog);    /* waves whose bounded wait for a ready word ran out since the program was loaded (0 unless something is broken) */
/* PROFILE_STRIDE n: with profiling on, only every n-th launch of a kind is bracketed by events */
/* FIR_ROWS: row tiles per wave of fir_tile, 0 = auto.  HOST_SPLIT: frames per piece of a host-pointer block (copies and kernels pipelined), 0 = whole block.
 * HOST_PIN 1: pin the caller's host buffers in place and remember them (the caller keeps them allocated until it sets 0 again) */
int avdsp_hip_prog_set_option(avdsp_hip_prog *prog, int key, int value);

/* the plugin's "tagoutput" (linux/avdsp_plugin.c:133-137) on one column of a device-resident S32 output block:
 * d_column = address of the column's sample in frame 0, stride in words; reset != 0 first sets the carried value */
int avdsp_hip_tag_output(avdsp_hip_prog *prog, void *d_column, int stride, int nframes, int reset, int reset_value, void *stream);

int avdsp_hip_tag_column_host(avdsp_hip_prog *prog, int *h_column, int nframes);      /* the same on a host copy of the column */

int avdsp_hip_synchronize(void *stream);
const char *avdsp_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* AVDSP_HIP_H_ */
