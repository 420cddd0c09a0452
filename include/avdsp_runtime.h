/*
 * avdsp_runtime.h -- C ABI of the MI355X-native AVDSP runtime (libavdsp_mi355x.so).
 *
 * Drop-in boundary: the five functions and the exported data below are exactly what the reference
 * runtime exports (module_avdsp/runtime/dsp_runtime.h:160-164, dsp_runtime.c:36-38,
 * dsp_header.c:10-86), so the reference hosts (linux/avdsp_plugin.c:126,178,316,328,336,
 * linux/dsprun.c:92,104,112,166, osx/dsprunosx.c:61,72,73,91) link against it unchanged.
 * The reference builds one library per DSP_FORMAT; this library carries the _2, _4 and _6 entry
 * points side by side (int64 / double+int samples / double+float samples).
 *
 * Everything that computes runs on the GPU (hand-written gfx950 kernels behind avdsp_hip.h).
 * There is NO CPU execution path in this library: a core that cannot be lowered to the device
 * kernels, a missing GPU, or a HIP error makes the call return a negative code and leaves a message
 * in dspRuntimeLastError().  (A CPU restatement exists under oracle/ -- test infrastructure only.)
 * Codes: -1 .. -6 the reference's (dsp_runtime.c:119-125,159-194: no header, no cores, checksum, opcode too new, buffer too
 * small; -1 / -2 from dspRuntimeReset: sample rate), -8 a core refused (no defined result / an offset outside the
 * buffer / a shape the call cannot take), -9 out of memory or table space, -10 a HIP error, -11 (sticky) a FIR wave of the overlap
 * mode gave up waiting for its cascade: the block is not valid, acknowledge with dspRuntimeReset() or
 * dspRuntimeSetOption("ready_timeouts", 0).
 *
 * Ownership is the reference's: the caller owns ONE contiguous int32 buffer holding the program
 * followed by the state ("data") area, dspRuntimeInit returns the program length so that
 * rundata = (int*)code + return value.  The device keeps a mirror of that buffer; state lives on
 * the device between blocks and is copied back by dspRuntimeSyncState() (checkpoint) or pushed by
 * dspRuntimeUploadState() (restore).
 */
#ifndef AVDSP_RUNTIME_H_
#define AVDSP_RUNTIME_H_

#include <stddef.h>

#include "avdsp_format.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- reference API (names, arguments, return codes unchanged) ---------------- */

/* dsp_runtime.c:42-59  -- numCore counts from 1; returns the DSP_CORE word (or the program start
 * when the program has no DSP_CORE and numCore... == first lookup), NULL when not found.          */
opcode_t *dspFindCore(opcode_t *codePtr, const int numCore);

/* dsp_runtime.c:62-77  -- skips CORE/NOP/PARAM/PARAM_NUM words in front of the executable part.   */
opcode_t *dspFindCoreBegin(opcode_t *ptr);

/* dsp_runtime.c:116-145 -- selects the sample-rate column, zeroes the state area (host mirror and
 * device), seeds dither.  0 ok, -1 unsupported fs, -2 fs outside the program's range.            */
int dspRuntimeReset(const int fs, int random, int defaultDither);

/* dsp_runtime.c:150-195 -- validates header, size, checksum, opcode level.  >= 0: program length in
 * words; -1 no header, -3 no cores, -4 checksum, -5 opcode too new, -6 buffer too small, or Reset's
 * code when fs != 0.  Extension: -7 when the program's encoding (header.format: 28 = Q28 integer,
 * 0 = float) would need dspChangeFormat for the entry point used later.          */
int dspRuntimeInit(opcode_t *codePtr, int maxSize, const int fs, int random, int defaultDither);

/* dsp_runtime.c:302-1314 -- one frame: samples[] is indexed by the program's IO numbers and is
 * read and written in place.  Returns 0 like the reference on success; negative on failure
 * (the reference cannot fail here; this implementation can: no GPU, core not lowerable).         */
int dspRuntime_2(opcode_t *core, int *rundata, int   *samples);   /* DSP_FORMAT_INT64        */
int dspRuntime_3(opcode_t *core, int *rundata, int   *samples);   /* DSP_FORMAT_FLOAT        */
int dspRuntime_4(opcode_t *core, int *rundata, int   *samples);   /* DSP_FORMAT_DOUBLE       */
int dspRuntime_5(opcode_t *core, int *rundata, float *samples);   /* DSP_FORMAT_FLOAT_FLOAT  */
int dspRuntime_6(opcode_t *core, int *rundata, float *samples);   /* DSP_FORMAT_DOUBLE_FLOAT */

extern dspHeader_t *dspHeaderPtr;           /* dsp_runtime.c:36 */
extern int          dspBiquadFreqSkip;      /* dsp_runtime.c:37 */
extern int          dspMantissa;            /* dsp_runtime.c:38 */
extern const char  *dspOpcodeText[DSP_MAX_OPCODE];                 /* dsp_header.c:10-73 */
long long dspQNM(double x, int n, int m);   /* dsp_header.c:75-77 */
long long dspQM64(double x, int m);         /* dsp_header.c:79-81 */
int       dspQM32(double x, int m);         /* dsp_header.c:83-85 */

/* ---------------- block extension (not in the reference) ----------------
 * Semantics: exactly nframes successive dspRuntime_N() calls over frame-interleaved buffers, i.e.
 * the host loop of linux/avdsp_plugin.c:98-141 for 32-bit samples:
 *     samples[in_io_base + k]  = in[n*in_stride + k]          k in [0, in_stride)
 *     dspRuntime_N(core, rundata, samples)
 *     out[n*out_stride + k]    = samples[out_io_base + k]     k in [0, out_stride)
 * Output slots the core never stores are left untouched.  in/out are HOST pointers here.
 * For the chain kernels the two windows must not overlap in IO numbers and must contain every IO the
 * core loads / stores.  The interpreter keeps a whole samples[] array per program, persistent like the
 * host's: slots outside the two windows are the program's own (values one core leaves for another, or
 * one frame for the next), and where the windows overlap the input is laid over the output.
 *
 * Two device paths sit behind these entry points.  A core that is a set of independent
 * LOAD|LOAD_GAIN -> BIQUADS* -> [FIR] -> [SAT0DB] -> STORE+ chains runs on parallel kernels in every
 * format: 2, 4 and 6 on the section-pipelined cascade (biquad_row / biquad_row_i64 / biquad_pipe) and
 * the MFMA FIR (fir_tile); 3 and 5 -- float accumulators and the truncating dspMulFloatFloat -- on
 * chain_rows (a lane per chain and section) and fir_lane (a lane per chain and frame), with
 * chain_lane for single frames and cascades longer than 16 sections.  Any other core -- X/Y
 * arithmetic, TPDF dither, delay lines, LOAD_MUX, RMS ... -- runs through the general device
 * interpreter (runs of identical strands on strand_lanes, a lane per strand).
 * Its frame-parallel kernel runs 64 frames of the block side by side (one
 * per lane, opcode by opcode) whenever the core hands nothing from one frame to the next except
 * opcode-private state (delay lines, filter state, meters ...), which is the case for every program
 * shipped with the reference; a core that does (a frame slot or memory read before it is written, a
 * TPDF_CALC behind its first use) runs frame by frame like the reference.  There is no CPU path. */
int dspRuntimeBlock_2(opcode_t *core, int *rundata, const int *in, int in_stride, int in_io_base,
                      int *out, int out_stride, int out_io_base, int nframes);
int dspRuntimeBlock_3(opcode_t *core, int *rundata, const int *in, int in_stride, int in_io_base,
                      int *out, int out_stride, int out_io_base, int nframes);
int dspRuntimeBlock_5(opcode_t *core, int *rundata, const float *in, int in_stride, int in_io_base,
                      float *out, int out_stride, int out_io_base, int nframes);
int dspRuntimeBlock_4(opcode_t *core, int *rundata, const int *in, int in_stride, int in_io_base,
                      int *out, int out_stride, int out_io_base, int nframes);
int dspRuntimeBlock_6(opcode_t *core, int *rundata, const float *in, int in_stride, int in_io_base,
                      float *out, int out_stride, int out_io_base, int nframes);

/* Same, with in/out resident in HBM (device pointers) and the work enqueued on `stream`
 * (a hipStream_t passed as void*, NULL = default stream).  Asynchronous: returns after enqueue. */
/* dspRuntimeBlock_N as a queue: submit returns once the block's copies and kernels are enqueued (its value: blocks in
 * flight, < 0 on error); up to four blocks are in flight, the copies of one under the kernels of another.  `in` and `out`
 * are pinned in place (hipHostRegister) and stay the library's -- allocated, unread, unwritten -- until
 * dspRuntimeBlockWait has let the block through; the registration ends there too, so a buffer may be freed as soon as
 * its block is back.  A host that cycles through the same few buffers sets dspRuntimeSetOption("host_pin", 1): the
 * registrations are then kept (registering 16 MB costs about as much as copying it) and the buffers must stay
 * allocated until "host_pin" is set back to 0, the program is released, or dspRuntimeRelease().
 * dspRuntimeBlockWait(m) returns when at most m submitted blocks are unfinished (oldest first; m = 0: all done) with
 * the number that still are.  Results are those of the same dspRuntimeBlock_N calls in the same order; every other
 * entry point waits for the queue by itself.  One core per block in flight: a program with several chain cores
 * submits core k+1 after dspRuntimeBlockWait(0) for core k (they share the output window), or uses dspRuntimeBlockAll. */
int dspRuntimeBlockSubmit(int format, opcode_t *core, int *rundata, const void *in, int in_stride, int in_io_base,
                          void *out, int out_stride, int out_io_base, int nframes);
int dspRuntimeBlockWait(int max_in_flight);

int dspRuntimeBlockDevice(int format, opcode_t *core, int *rundata,
                          const void *d_in, int in_stride, int in_io_base,
                          void *d_out, int out_stride, int out_io_base, int nframes, void *stream);

/* All cores of the loaded program over one block: the result of dspRuntimeBlock_N for core 1, 2, ... in turn
 * (the host loop of linux/avdsp_plugin.c:95-142 runs cores outermost too), in one call.  The samples cross PCIe
 * once instead of once per core, and cores that do not meet -- no frame slot, memory word (STORE_MEM, LOAD_MUX /
 * TPDF result), state range or dither global written by one and touched by the other -- run at the same time
 * on the device; cores that do are kept in program order.  Interpreted cores are cut further, into groups of
 * strands (LOAD ... STORE runs) that hand nothing to each other, which run side by side as well ("strand_split",
 * default 1).  dspRuntimeGetOption("levels") / ("cores") / ("pieces") tell how the latest call was arranged.  format = DSP_FORMAT 2..6;
 * in/out: int32 or float samples as in dspRuntimeBlock_N (host pointers) / device pointers + stream.       */
/* N instances of the loaded program side by side (extension; avdsp_host.c has the semantics): the reference is one program on one
 * core per process -- a host with many independent streams of the same program (a thousand stereo crossovers) loads it once.
 * Instance i: sample blocks at d_in + i * in_inst_words and d_out + i * out_inst_words (32-bit words), state of its own, copied
 * from the program's at the first block call after dspRuntimeSetInstances.  dspRuntimeInstanceState(i, dst) brings back instance
 * i's data area (dataSize words).  Cores for the frame-parallel interpreter (the reference's own
 * programs): an instance is a further copy of the device state (round 4); a program of CHAIN cores only (round 5): an instance is a further block of
 * chains in the chain kernels' launches -- 8 channels x 8 biquads in 512 instances are the 4096 rows of one cascade launch, 78 Gsamples/s
 * at 256-frame blocks -- with state, FIR history and parameters of its own in a copy of the mirror).  While a chain program runs as
 * instances the ordinary block calls on it are refused; dspRuntimeSetInstances(0) ends it (the single program continues from instance
 * 0's state, "strand_lanes" is what it was).  A program with cores of both kinds runs every core on the interpreter while it has
 * instances ("generic" reads 1 meanwhile; dspRuntimeSetInstances(0) gives it its chain plans back). */
int dspRuntimeSetInstances(int n);
int dspRuntimeBlockAllInstancesDevice(int format, int *rundata, const void *d_in, int in_stride, int in_io_base, size_t in_inst_words,
                                      void *d_out, int out_stride, int out_io_base, size_t out_inst_words, int nframes, void *stream);
int dspRuntimeInstanceState(int inst, int *dst);

int dspRuntimeBlockAll(int format, int *rundata, const void *in, int in_stride, int in_io_base,
                       void *out, int out_stride, int out_io_base, int nframes);
int dspRuntimeBlockAllDevice(int format, int *rundata, const void *d_in, int in_stride, int in_io_base,
                             void *d_out, int out_stride, int out_io_base, int nframes, void *stream);
/* the same with the plugin's packed PCM input (AVDSP_PCM_S16 / S24_3LE / S32, formats 2, 3, 4), unpacked once */
int dspRuntimeBlockAllPcm(int format, int *rundata, int pcm, const void *src, int in_stride, int in_io_base,
                          int *dst, int out_stride, int out_io_base, int nframes);

/* The host's sample-format step (linux/avdsp_plugin.c:103-121): `src` holds packed little-endian PCM,
 * frame-interleaved [nframes][in_stride]; pcm = AVDSP_PCM_S32 | AVDSP_PCM_S24_3LE | AVDSP_PCM_S16
 * (include/avdsp_hip.h).  The unpacking to s.31 words happens on the device; output is S32 like the
 * plugin's.  format = 2, 3 or 4 (the int-sample models).  dspRuntimeUnpackPcmDevice is the same
 * conversion alone, for producers whose PCM already sits in HBM (d_dst 16-byte aligned).          */
int dspRuntimeBlockPcm(int format, opcode_t *core, int *rundata, int pcm, const void *src, int in_stride, int in_io_base,
                       int *dst, int out_stride, int out_io_base, int nframes);
int dspRuntimeUnpackPcmDevice(int pcm, const void *d_src, int *d_dst, long long nsamples, void *stream);

/* The plugin's "tagoutput" option (linux/avdsp_plugin.c:133-137, key :262): bits 8..15 of the first output channel of a core
 * carry a count derived from the previous sample's upper half, so that a bit-perfect transport can be verified downstream:
 *     sample' = (sample & 0xFFFF0000) | (previoussample & 0x0000FF00);   previoussample = ((sample & 0xFFFF0000) >> 8) + 0x100
 * Call it after a core's block, on that core's first output column (column = its IO number - out_io_base), cores in program
 * order as the plugin's loop runs them; the carried value lives on the device.  Device-resident block, or host block.  */
int dspRuntimeTagOutputDevice(void *d_out, int out_stride, int column, int nframes, void *stream);
int dspRuntimeTagOutput(int *out, int out_stride, int column, int nframes);
int dspRuntimeTagOutputReset(int previoussample);

/* device state -> rundata (the buffer stays the checkpoint) / rundata -> device state (restore).
 * Sync also brings back the program words (DSP_STORE_MEM writes into the program's parameter
 * section, dsp_runtime.c:755-760).                                                              */
int dspRuntimeSyncState(int *rundata);
int dspRuntimeUploadState(const int *rundata);
/* The host edited parameters (gains, biquad coefficients, delay values, bypass flags, taps) in the program
 * words, as the reference allows between any two frames: carry the edits to the device.  State is kept. */
int dspRuntimeUploadParams(void);

/* ---------------- channel sharding across the GPUs of a node (not in the reference) ----------------
 * The reference has one notion of several executors: DSP_CORE segments meant for parallel XMOS threads, which its
 * Linux hosts run one after the other (linux/avdsp_plugin.c:95).  Here the unit is finer: the chains of a chain core
 * are independent channels (the lowering proves it: no IO stored twice, none both loaded and stored), so
 * `world` processes -- one per GPU -- each run a contiguous, balanced range of them and never exchange anything.
 *
 * dspRuntimeSetShard(rank, world): from now on every chain core of the loaded program is lowered to the chains
 * [lo, hi) of rank `rank` only, lo = rank*q + min(rank, r), hi = lo + q + (rank < r), (q, r) = divmod(chains, world).
 * It works on any loaded .bin (the cut is made after lowering, on the chain list), keeps the device state
 * (FIR histories return to the mirror and are picked up again), and (0, 1) switches it off.  The block calls'
 * windows then only need to cover the IO numbers of the rank's own chains: a host hands over its column slice
 *     in  = x + in_io_min-th column,  in_io_base  = in_io_min,  in_stride  = whatever pitch the slice has
 * and receives its slice of the output columns; dspRuntimeShardInfo() reports those IO ranges (host-only, nothing
 * runs).  Cores that are not chain cores (interpreter) are not cut: every rank runs them whole (replicas).
 * dspRuntimeSyncState() returns this rank's view: its own chains' state advanced, the other chains' untouched. */
int dspRuntimeSetShard(int rank, int world);
int dspRuntimeShardInfo(int format, opcode_t *core, int *total_chains, int *first_chain, int *nchains,
                        int *in_io_min, int *in_io_max, int *out_io_min, int *out_io_max);

/* Host-only: the tail of an interpreted core that is N >= 2 repetitions of one opcode sequence -- one strand per channel, the shape of
 * the reference's crossover programs -- runs with lane = strand (avdsp_hip.h, strand plans).  strands = 0: the core has no such tail, or
 * the run is too short for the current "strand_lanes" setting (default: more than 64 strands; 2: any run). */
int dspRuntimeStrandInfo(int format, opcode_t *core, int *strands, int *ops_per_strand, int *prefix_words);

/* Tunables: "fir_impl" 0 = plain tap loop, 1 = MFMA (fir_tile, default), 2 / 3 / 4 = the other MFMA kernels kept for comparison (fir_mfma,
 * fir_stream, fir_flow: same results); "biquad_impl" 0 = lane per channel,
 * 1 = section-pipelined (default: biquad_row where it applies, else biquad_pipe), 2 = round 2's biquad_pipe throughout; "interp_impl" 0 = interpreter always frame by frame, 1 = frame-parallel
 * where the core allows it (default); "strand_split" 0 = dspRuntimeBlockAll keeps cores whole; "strand_lanes" 0 = strand runs stay with the interpreter, 1 = runs of more than 64 strands on lanes (default: up to 64 strand groups get a wave each from the interpreter, which is faster), 2 = every run; "generic" 1 = every core through the interpreter;
 * "device" = HIP device ordinal (before the first block);
 * "overlap" (chain cores with cascades in front of FIRs, blocks resident on the device) 1 = the cascades run up to three blocks ahead of
 * the FIRs on a stream of the library's own -- the caller then guarantees that a block's INPUT is complete in memory when the call is made
 * (stream order no longer covers it); 2 = also the FIRs of consecutive blocks on two streams in turn, so that one starts while the other's
 * last workgroups leave (worth 3.5 % on a 4096-channel program, a loss below ~2000 channels) -- the caller then also guarantees that the
 * OUTPUT block of a call is not one an earlier call's FIR may still be writing (the caller's stream still waits for every block's end).
 * Results are identical in every mode.  Under "overlap": "ring_wait" 1 (default) = the HOST waits (at most 1 ms, then it leaves it to the
 * stream after all) until the FIR three blocks back has ended before it enqueues a block's cascade -- the call then returns no more than
 * three blocks ahead of the device, and the cascades' stream carries no wait packet (worth 10 % on a 512-channel shard); 0 = that stream
 * waits.  "ready_words" = how a block's FIR finds its cascades' block: 0 an event between the two queues, 1 words published by the
 * cascade's waves and polled by the FIR's (slower), 2 words set by a kernel behind the cascade (no wait packet on the FIRs' stream),
 * -1 (default) = 2 where the FIR is the bound, else 0.  Round 5: ready words are only taken once kernels of the FIRs' stream and of the
 * cascades' stream have been SEEN to run side by side (a one-time probe per stream; dspRuntimeGetOption("side_by_side") 1 / 0 / -1 not
 * tried yet), and the cascades' stream is made anew while it shares a hardware queue with the FIRs' ("streams_remade" says how often) --
 * without that the mode does not overlap anything.  A FIR wave whose bounded wait for a ready word runs out all the same (a GPU that stops
 * running two queues at once) makes EVERY later call fail with -11 until dspRuntimeReset() or dspRuntimeSetOption("ready_timeouts", 0)
 * acknowledges; dspRuntimeGetOption("ready_timeouts") counts such waves, ("ready_mode") tells what the latest launch used.
 * "cu_split" k (experiment; 0 = off): under "overlap" the cascades' stream on k CUs of its own (8, 16, 32 ...; a CU mask),
 * the FIRs on a stream of the library's with the complementary mask (negative k: only the cascades masked); hand over a non-blocking
 * stream of your own with it -- CU-masked streams are blocking streams, beside the null stream every launch on them synchronises.
 * "group_fanout" 1 (default) = a chain core's cascades of up to 16 sections run as ONE launch whatever their lengths, longer ones side by
 * side over up to four streams; 0 = one launch per section count, one after the other (as through round 4; results identical).
 * A program's options are also the defaults of programs loaded later.                                */
int dspRuntimeSetOption(const char *key, int value);
int dspRuntimeGetOption(const char *key);

/* Kernel timing with HIP events on the launch stream: enable with dspRuntimeSetOption("profile", 1), or
 * ("profile", 2 * mask) to time only the kinds whose bit is set in mask (each event pair costs the stream a few
 * microseconds);
 * kind 0 = biquad cascade, 1 = FIR, 2 = pass-through, 3 = general interpreter frame by frame,
 * 4 = PCM unpack, 5 = general interpreter frame-parallel.  Returns the summed duration (ms) and launch
 * count of the launches recorded since the previous read.                                        */
int dspRuntimeKernelTime(int kind, double *total_ms, int *launches);

/* Introspection of the lowered core (what the device plan contains), host-only.  nchains > 0: the
 * parallel chain kernels take it; nchains == 0: the general interpreter takes it; negative: neither
 * (encoding mismatch, an offset outside the buffer, an opcode with no defined result in this format). */
int dspRuntimeCoreInfo(int format, opcode_t *core, int *nchains, int *max_sections, int *max_taps);

const char *dspRuntimeLastError(void);
void        dspRuntimeRelease(void);        /* frees device memory of every loaded program; the next Init starts clean */
/* Several programs in one process.  dspRuntimeInit(codePtr, ...) loads a program into a context of its own, keyed by the
 * caller's buffer (loading the same buffer again restarts that program); every call that takes a pointer into a program
 * -- a core, its data area -- addresses that program, and the reference's exported data (dspHeaderPtr, dspBiquadFreqSkip,
 * dspMantissa) follow.  Calls without one (dspRuntimeReset, options, shard, timers, tagoutput, wait) address the program
 * of the latest call that named one, or dspRuntimeSelect's.  Each program has its own device copy, GPU ("device"
 * option at the time of its first block call), shard and plans; an option set while a program is current also becomes
 * the default of programs loaded later.  One process can so drive several GPUs (one buffer per GPU, each with its
 * dspRuntimeSetShard), or several programs on one.  Up to 64.                                                    */
int         dspRuntimeSelect(const void *ptr_into_program);
int         dspRuntimeReleaseProgram(opcode_t *codePtr);

#ifdef __cplusplus
}
#endif
#endif /* AVDSP_RUNTIME_H_ */
