/*
 * avdsp_encoder.c -- the program encoder behind include/avdsp_encoder.h, plain C99, host only.
 *
 * What it restates (behaviour, not text): module_avdsp/encoder/dsp_encoder.c (opcode emission,
 * PARAM sections, data-area allocation, length back-patching, header/checksum at END_OF_CODE),
 * encoder/dsp_filters.c (biquad design formulas and the Bessel / Butterworth / Linkwitz-Riley presets)
 * and dspCreateBuffer/dspReadBuffer of encoder/dsp_fileaccess.c.  The output is held byte-identical to
 * the reference encoder's by tests/test_encoder.py.
 *
 * Structure: one `struct emitter` (the reference keeps ~30 file-scope statics).  An opcode whose
 * payload length is not known when its head word is written stays "open" until the next opcode is
 * started (`begin_opcode`), which back-patches its skip.  A PARAM / PARAM_NUM region stays current
 * until the next opcode; inside it at most one counted *section* (biquad bank, mux list, FIR impulses)
 * is being filled.
 */
#include "avdsp_encoder.h"

#include <stdlib.h>
#include <string.h>

#define ENCODER_VERSION ((1 << 8) | (0 << 4) | 2)            /* dsp_encoder.c:12 */

dspHeader_t *dspHeaderPtr;                                     /* shared name with the runtime library */
int dspMinSamplingFreq = DSP_DEFAULT_MIN_FREQ;
int dspMaxSamplingFreq = DSP_DEFAULT_MAX_FREQ;

static struct emitter {
    opcode_t *w;                 /* the caller's table */
    dspHeader_t *hdr;            /* its first 12 words (the exported dspHeaderPtr may be shared with the runtime) */
    int cap, n;                  /* capacity, next free word */
    int data;                    /* next free word of the data (state) area */
    int open_op;                 /* head word still waiting for its skip, -1 = none */
    int region;                  /* head word of the current PARAM / PARAM_NUM, 0 = not inside one */
    int want_at, want_words;     /* "at least this many words must follow here" (inline parameters) */
    struct { int opcode, left, count, first; } sect;   /* counted section being filled */
    int core_at;                 /* CORE head whose IO bitmaps are still to be written */
    unsigned in_all, out_all, in_core, out_core;
    int max_opcode;
    /* The header's maxOpcode is the maximum over "the word where the previous opcode was believed to
     * start" (dsp_encoder.c:296-297), a position the reference's listing code also moves; these two
     * cursors follow the same rules so the header matches in every case.                           */
    int list_op, list_mark;
    int format, mant, io_max, nf;
} E = { .open_op = -1 };

static void fatal(const char *msg)
{
    fprintf(stderr, "FATAL ERROR : %s\n", msg);              /* dsp_encoder.c:58-61 */
    exit(1);
}

/* ---------------- words ---------------- */
int opcodeIndex(void) { return E.n; }

static int reserve(int words)
{
    int at = E.n;
    if (at + words > E.cap) fatal("YOUR DSP CODE IS TOO LARGE FOR THE ARRAY PROVIDED");
    E.n += words;
    return at;
}

int addCode(int code)
{
    int at = E.n;
    E.w[at].i32 = code;
    reserve(1);
    E.w[E.n].i32 = DSP_END_OF_CODE;                          /* the table always ends in a terminator */
    return at;
}

int addFloat(float value)
{
    opcode_t v;
    v.f32 = value;
    return addCode(v.i32);
}

static int put_head(int opcode, int low16) { return addCode((opcode << 16) | (low16 & 0xFFFF)); }

/* a word holding `target - base`, or the distance to the word right behind this one when target == 0 */
static int put_offset(int target, int base)
{
    return addCode(target ? target - base : E.n + 1 - base);
}

static int pad_to_even(void) { if (E.n & 1) addCode(0); return E.n; }     /* next word 8-byte aligned */
static int pad_to_odd(void)  { if (!(E.n & 1)) addCode(0); return E.n; }

/* ---------------- data area ---------------- */
static int take_data(int words)
{
    int at = E.data;
    addCode(at);
    E.data += words;
    return at;
}
static int take_data_even(int words) { if (E.data & 1) E.data++; return take_data(words); }
static int take_data_odd(int words)  { if (!(E.data & 1)) E.data++; return take_data(words); }

/* ---------------- listing cursors (see struct emitter) ---------------- */
static void list_mark_here(void) { E.list_mark = E.n; }
static void list_flush(void)
{
    if (!(E.list_mark < E.list_op)) E.list_op = E.list_mark;
    E.list_mark = E.n;
}

/* ---------------- sections inside a PARAM region ---------------- */
static void need_region(void)
{
    if (!E.region) fatal("Currently not in a PARAM or PARAM_NUM space.");
}

static void close_section(void)                               /* dsp_encoder.c:186-215 */
{
    if (!E.sect.opcode) return;
    if (E.sect.left > 0) fatal("Section already started and not finished.");
    opcode_t *first = &E.w[E.sect.first];
    switch (first->op.opcode) {
    case DSP_BIQUADS:
        first->s16.low = (short)E.sect.count;
        E.sect.opcode = 0;
        list_mark_here();
        break;
    case DSP_LOAD_MUX:
        list_flush();
        first->s16.low = (short)E.sect.count;
        E.sect.opcode = 0;
        break;
    case DSP_FIR:
        if (E.sect.count != E.nf) fatal("Missing impulse in the fir param section.");
        break;
    default: break;
    }
}

static int open_section(int opcode, int expected)             /* :218-227 */
{
    close_section();
    list_flush();
    need_region();
    E.sect.opcode = opcode;
    E.sect.left = expected;
    E.sect.count = 0;
    E.sect.first = E.n;
    return E.n;
}

/* one more entry of the open section; returns 0 when that was the last expected one */
static int section_entry(int opcode)                          /* :229-256 */
{
    need_region();
    if (!E.sect.opcode) fatal("No section defined or started.");
    if (opcode && E.sect.opcode != opcode) fatal("Section already started for another opcode.");
    E.sect.count++;
    if (E.sect.left > 0) {
        if (--E.sect.left == 0) E.sect.opcode = 0;
    } else if (E.sect.left < 0 && E.sect.count > -E.sect.left)
        fatal("too much parameters in this section.");
    return E.sect.opcode;
}

/* ---------------- opcode framing ---------------- */
static void begin_opcode(void)                                /* dsp_encoder.c:273-301 */
{
    if (!E.w) fatal("dspEncoderInit has not been launched first.");
    if (E.region) { close_section(); E.region = 0; }
    if (E.want_at) {
        if (E.n - E.want_at < E.want_words) fatal("not enough parameters provided below this opcode.");
        E.want_at = 0; E.want_words = 0;
    }
    if (E.open_op != -1) {
        E.w[E.open_op].op.skip = (unsigned short)(E.n - E.open_op);
        E.open_op = -1;
    }
    int seen = E.w[E.list_op].op.opcode;
    if (seen > E.max_opcode) E.max_opcode = seen;
    list_flush();
    E.list_op = E.n;
}

static int op_single(int opcode) { begin_opcode(); return put_head(opcode, 1); }
static int op_open(int opcode)   { begin_opcode(); E.open_op = put_head(opcode, 0); return E.open_op; }

static void want_below(int words) { E.want_at = E.n; E.want_words = words; }
static void want_below_if0(int paramAddr, int words) { if (!paramAddr) want_below(words); }

/* ---------------- PARAM address checks: dsp_encoder.c:391-452 ---------------- */
static int find_region(int index, int size)
{
    int last = index + size - 1, pos = 0;
    for (;;) {
        int code = E.w[pos].op.opcode, skip = E.w[pos].op.skip;
        int first = code == DSP_PARAM ? 1 : code == DSP_PARAM_NUM ? 2 : 0;
        if (first) {
            int begin = pos + first, end = skip ? pos + skip : E.n;
            if (index >= begin && index < end) {
                if (last < end) return (begin << 16) | end;
                fatal("memory space expected is too large for this PARAM or PARAM_NUM.");
            }
        }
        if (!skip) fatal("Index provided not found in any PARAM or PARAM_NUM space.");
        pos += skip;
    }
}

static int find_region_of(int index, int size, int opcode)
{
    if (opcode && E.w[index].op.opcode != opcode)
        fatal("the parameter adress is not pointing on a proper section of data.");
    return find_region(index, size);
}

static void check_io(int io)
{
    if (io < 0 || io >= E.io_max) fatal("IO out of range.");
}
static void check_range(int v, int lo, int hi)
{
    if (v < lo || v > hi) fatal("value not in expected range");
}
static void mark_in(int io)  { if (io < 32) { E.in_all  |= 1u << io; E.in_core  |= 1u << io; } }
static void mark_out(int io) { if (io < 32) { E.out_all |= 1u << io; E.out_core |= 1u << io; } }

static void flush_core_io(void)                               /* :454-462 */
{
    if (E.core_at) {
        E.w[E.core_at + 1].u32 = E.in_core;
        E.w[E.core_at + 2].u32 = E.out_core;
        E.core_at = 0;
    }
}

/* a gain / value parameter: Q4.28 when the program is integer-encoded, float otherwise (:608-618) */
static int put_param(dspGainParam_t v)
{
    if (E.format < DSP_FORMAT_FLOAT) return addCode(DSP_QM32(v, E.mant));
    return addFloat(v);
}

/* ---------------- init / header ---------------- */
void setSerialHash(unsigned hash) { E.hdr->serialHash = hash; }

void dspEncoderFormat(int format)                             /* :308-331 */
{
    if (format > DSP_FORMAT_DOUBLE_FLOAT) { E.format = DSP_FORMAT_INT64; E.mant = format; }
    else if (format == 0)                 { E.format = DSP_FORMAT_FLOAT; E.mant = 0; }
    else                                  { E.format = format; E.mant = DSP_MANT; }
    E.hdr->format = (unsigned short)(E.format < DSP_FORMAT_FLOAT ? E.mant : 0);
}

void dspEncoderInit(opcode_t *opcodeTable, int max, int format, int minFreq, int maxFreq, int maxIO)
{
    memset(&E, 0, sizeof E);
    E.w = opcodeTable; E.cap = max; E.open_op = -1;
    E.hdr = dspHeaderPtr = (dspHeader_t *)opcodeTable;
    dspMinSamplingFreq = minFreq; dspMaxSamplingFreq = maxFreq;
    E.nf = maxFreq - minFreq + 1;
    E.io_max = maxIO;
    E.open_op = put_head(DSP_HEADER, 0);
    reserve(AVDSP_HEADER_WORDS - 1);
    E.hdr->totalLength = 0; E.hdr->dataSize = 0; E.hdr->checkSum = 0;
    E.hdr->numCores = 0;
    E.hdr->version = ENCODER_VERSION;
    dspEncoderFormat(format);
    E.hdr->maxOpcode = DSP_MAX_OPCODE - 1;
    E.hdr->freqMin = minFreq; E.hdr->freqMax = maxFreq;
    E.hdr->usedInputs = 0; E.hdr->usedOutputs = 0;
    setSerialHash(0);
}

int dsp_END_OF_CODE(void)                                     /* :509-551 */
{
    flush_core_io();
    begin_opcode();
    put_head(DSP_END_OF_CODE, 0);
    if (E.n & 1) addCode(0);
    begin_opcode();
    E.hdr->totalLength = E.n;
    E.hdr->dataSize = E.data;
    unsigned sum; int cores;
    avdspChecksumWalk(E.w, (unsigned)E.n, &sum, &cores);
    E.hdr->checkSum = sum;
    E.hdr->numCores = cores ? cores : 1;
    E.hdr->maxOpcode = (unsigned short)E.max_opcode;
    E.hdr->usedInputs = E.in_all;
    E.hdr->usedOutputs = E.out_all;
    return E.n;
}

/* dump files are host tooling outside this library; the calls keep their side effect on the cursors */
static int locate_param(int addr)                             /* :391-416 */
{
    int pos = 0, num = 0;
    for (;;) {
        int code = E.w[pos].op.opcode, skip = E.w[pos].op.skip, first = 0;
        if (code == DSP_PARAM || code == DSP_HEADER) { first = 1; num = 0; }
        if (code == DSP_PARAM_NUM) { first = 2; num = E.w[pos + 1].i32; }
        if (first) {
            int begin = pos + first, end = skip ? pos + skip - 1 : E.n - 1;
            if (addr >= begin && addr <= end) return num ? ((addr - begin) | (num << 16)) : addr;
        }
        if (!skip) fatal("Index provided not found in any PARAM or PARAM_NUM space.");
        pos += skip;
    }
}
void dsp_dumpParameter(int addr, int size, char *name) { (void)size; (void)name; locate_param(addr); list_flush(); }
void dsp_dumpParameterNum(int addr, int size, char *name, int num) { (void)num; dsp_dumpParameter(addr, size, name); }

int opcodeIndexAligned8(void)    { if (E.n & 1) op_single(DSP_NOP); return E.n; }
int opcodeIndexMisAligned8(void) { if (!(E.n & 1)) op_single(DSP_NOP); return E.n; }

/* ---------------- opcodes without payload: :621-690 ---------------- */
void dsp_NOP(void)    { op_single(DSP_NOP); }
void dsp_CLRXY(void)  { op_single(DSP_CLRXY); }
void dsp_SWAPXY(void) { op_single(DSP_SWAPXY); }
void dsp_COPYXY(void) { op_single(DSP_COPYXY); }
void dsp_COPYYX(void) { op_single(DSP_COPYYX); }
void dsp_ADDXY(void)  { op_single(DSP_ADDXY); }
void dsp_ADDYX(void)  { op_single(DSP_ADDYX); }
void dsp_SUBXY(void)  { op_single(DSP_SUBXY); }
void dsp_SUBYX(void)  { op_single(DSP_SUBYX); }
void dsp_MULXY(void)  { op_single(DSP_MULXY); }
void dsp_DIVXY(void)  { op_single(DSP_DIVXY); }
void dsp_DIVYX(void)  { op_single(DSP_DIVYX); }
void dsp_AVGXY(void)  { op_single(DSP_AVGXY); }
void dsp_AVGYX(void)  { op_single(DSP_AVGYX); }
void dsp_SQRTX(void)  { op_single(DSP_SQRTX); }
void dsp_NEGX(void)   { op_single(DSP_NEGX); }
void dsp_NEGY(void)   { op_single(DSP_NEGY); }
void dsp_WHITE(void)  { op_single(DSP_WHITE); }
void dsp_SAT0DB(void) { op_single(DSP_SAT0DB); }
void dsp_SAT0DB_TPDF(void) { op_single(DSP_SAT0DB_TPDF); }

void dsp_CORE(void)                                           /* :624-632 */
{
    flush_core_io();
    E.in_core = E.out_core = 0;
    E.core_at = op_open(DSP_CORE);
    int at = reserve(2);
    E.w[at].i32 = 0; E.w[at + 1].i32 = 0;
}

void dsp_SERIAL(unsigned hash) { op_open(DSP_SERIAL); addCode((int)hash); }

/* ---------------- saturation with gain, dither source, shift: :692-746 ---------------- */
static void sat_gain(int paramAddr, int opcode)
{
    int at = op_open(opcode);
    if (paramAddr) find_region(paramAddr, 1);
    put_offset(paramAddr, at);
    want_below_if0(paramAddr, 1);
}
void dsp_SAT0DB_TPDF_GAIN(int paramAddr) { sat_gain(paramAddr, DSP_SAT0DB_TPDF_GAIN); }
void dsp_SAT0DB_GAIN(int paramAddr)      { sat_gain(paramAddr, DSP_SAT0DB_GAIN); }
void dsp_SAT0DB_GAIN_Fixed(dspGainParam_t gain)      { sat_gain(0, DSP_SAT0DB_GAIN); put_param(gain); }
void dsp_SAT0DB_TPDF_GAIN_Fixed(dspGainParam_t gain) { sat_gain(0, DSP_SAT0DB_TPDF_GAIN); put_param(gain); }

static int tpdf(int opcode, int bits)
{
    op_open(opcode);
    check_range(bits, 0, 32);
    addCode(bits);
    return take_data_even(2);
}
int dsp_TPDF_CALC(int bits) { return tpdf(DSP_TPDF_CALC, bits); }
int dsp_TPDF(int bits)      { return tpdf(DSP_TPDF, bits); }

void dsp_SHIFT(int bits) { op_open(DSP_SHIFT); addCode(bits); }
void dsp_SHIFT_FixedInt(int bits) { dsp_SHIFT(bits); }

/* ---------------- sample IO: :756-847, 1027-1043 ---------------- */
void dsp_LOAD(int IO)
{
    check_io(IO);
    mark_in(IO);
    op_open(DSP_LOAD);
    addCode(IO);
}

void dsp_LOAD_GAIN(int IO, int paramAddr)
{
    int at = op_open(DSP_LOAD_GAIN);
    check_io(IO);
    mark_in(IO);
    addCode(IO);
    if (paramAddr) find_region(paramAddr, 1);
    put_offset(paramAddr, at);
    want_below_if0(paramAddr, 1);
}

void dsp_LOAD_GAIN_Fixed(int IO, dspGainParam_t gain) { dsp_LOAD_GAIN(IO, 0); put_param(gain); }

int dsp_LOAD_MUX(int paramAddr)
{
    int at = op_open(DSP_LOAD_MUX);
    find_region_of(paramAddr, 2, DSP_LOAD_MUX);
    put_offset(paramAddr, at);
    return take_data_even(2);
}

int dspLoadMux_Inputs(int number)
{
    open_section(DSP_LOAD_MUX, number);
    return put_head(DSP_LOAD_MUX, number);
}

void dspLoadMux_Data(int in, dspGainParam_t gain)
{
    check_io(in);
    mark_in(in);
    int more = section_entry(DSP_LOAD_MUX);
    addCode(in);
    put_param(gain);
    if (!more) list_mark_here();
}

void dsp_STORE(int IO)
{
    check_io(IO);
    op_open(DSP_STORE);
    addCode(IO);
    mark_out(IO);
}

void dsp_LOAD_STORE(void) { op_open(DSP_LOAD_STORE); want_below(2); }

void dspLoadStore_Data(int in, int out)
{
    if (!E.want_at) fatal("no parameter expected here.");
    if (E.w[E.want_at - 1].op.opcode != DSP_LOAD_STORE) fatal("incompatible with the previous opcode generated.");
    check_io(in); check_io(out);
    addCode(in); addCode(out);
    mark_in(in); mark_out(out);
}

/* ---------------- PARAM regions and plain data: :849-867, 878-885, 916-923, 966-1024 ---------------- */
int dsp_PARAM(void) { E.region = op_open(DSP_PARAM); return E.region; }

int dsp_PARAM_NUM(int num)
{
    int at = op_open(DSP_PARAM_NUM);
    E.region = at;
    addCode(num);
    return at;
}

static void need_plain_region(void) { need_region(); close_section(); }

int dspGain_Default(dspGainParam_t gain)
{
    need_plain_region();
    int at = put_param(gain);
    E.list_op = E.n;
    return at;
}
int dspValue_Default(float value) { return dspGain_Default(value); }

int dspDataTableInt(int *data, int n)
{
    need_plain_region();
    int at = E.n;
    for (int i = 0; i < n; i++) addCode(data[i]);
    E.list_mark = E.n;
    return at;
}

int dspDataTableFloat(float *data, int n)
{
    list_flush();
    need_plain_region();
    int at = E.n;
    for (int i = 0; i < n; i++) put_param(data[i]);
    list_mark_here();
    return at;
}

static int data_words(const int *v, int n)
{
    need_plain_region();
    int at = E.n;
    for (int i = 0; i < n; i++) addCode(v[i]);
    E.list_op = E.n;
    return at;
}
int dspData2(int a, int b) { int v[2] = { a, b }; return data_words(v, 2); }
int dspData4(int a, int b, int c, int d) { int v[4] = { a, b, c, d }; return data_words(v, 4); }
int dspData6(int a, int b, int c, int d, int e, int f) { int v[6] = { a, b, c, d, e, f }; return data_words(v, 6); }
int dspData8(int a, int b, int c, int d, int e, int f, int g, int h) { int v[8] = { a, b, c, d, e, f, g, h }; return data_words(v, 8); }

/* The argument is formed as (2 pi i) * (1 / samples): the reference's source divides, its build flags
 * (-Ofast) turn that into a multiplication by the reciprocal, and a handful of table entries differ by
 * one LSB between the two.  The reference's binaries (osx .bin files, dspcreate) are the target.  */
int dspGenerator_Sine(int samples)                            /* :1191-1203 */
{
    need_plain_region();
    int at = E.n;
    check_range(samples, 4, 1024);
    for (int i = 0; i < samples; i++)
        addCode(DSP_QM32(sin((2.0 * M_PI * (double)i) * (1.0 / (double)samples)), 31));
    list_mark_here();
    return at;
}

/* ---------------- gains, immediates: :870-959 ---------------- */
void dsp_GAIN(int paramAddr)
{
    int at = op_open(DSP_GAIN);
    if (paramAddr) find_region(paramAddr, 1);
    put_offset(paramAddr, at);
    want_below_if0(paramAddr, 1);
}
void dsp_GAIN_Fixed(dspGainParam_t gain) { int at = op_open(DSP_GAIN); put_offset(0, at); put_param(gain); }
void dsp_VALUE_Fixed(float value)        { int at = op_open(DSP_VALUE); put_offset(0, at); put_param(value); }
void dsp_VALUE_FixedInt(int value)       { int at = op_open(DSP_VALUE_INT); put_offset(0, at); addCode(value); }
void dsp_VALUE(int paramAddr)
{
    int at = op_open(DSP_VALUE);
    find_region(paramAddr, 1);
    put_offset(paramAddr, at);
}
void dsp_DIV_Fixed(float value)  { op_open(DSP_DIV_VALUE); put_param(value); }
void dsp_DIV_FixedInt(int value) { op_open(DSP_DIV_VALUE_INT); addCode(value); }
void dsp_MUL_Fixed(float value)  { op_open(DSP_MUL_VALUE); put_param(value); }
void dsp_MUL_FixedInt(int value) { op_open(DSP_MUL_VALUE_INT); addCode(value); }
void dsp_AND_FixedInt(int value) { op_open(DSP_AND_VALUE_INT); addCode(value); }
void dsp_DELAY_1(void)           { op_open(DSP_DELAY_1); take_data_even(2); }

/* ---------------- memories: :1045-1086 ---------------- */
static void mem_op(int opcode, int paramAddr, int index)
{
    int at = op_open(opcode);
    int where = paramAddr + index * 2;
    find_region(where, 2);
    put_offset(where, at);
}
void dsp_LOAD_MEM_Index(int paramAddr, int index)  { mem_op(DSP_LOAD_MEM, paramAddr, index); }
void dsp_STORE_MEM_Index(int paramAddr, int index) { mem_op(DSP_STORE_MEM, paramAddr, index); }
void dsp_LOAD_MEM(int paramAddr)  { dsp_LOAD_MEM_Index(paramAddr, 0); }
void dsp_STORE_MEM(int paramAddr) { dsp_STORE_MEM_Index(paramAddr, 0); }

int dspMem_LocationMultiple(int number)
{
    close_section();
    need_region();
    pad_to_even();
    int at = reserve(2 * number);
    memset(&E.w[at], 0, (size_t)(2 * number) * sizeof(opcode_t));
    return at;
}
int dspMem_Location(void) { return dspMem_LocationMultiple(1); }

/* ---------------- delay lines: :1088-1175 ---------------- */
static void delay_param(int paramAddr, int opcode)
{
    find_region(paramAddr, 1);
    int at = op_open(opcode);
    int size = E.w[paramAddr].s16.high;                       /* line length in samples at the highest rate */
    addCode(size);
    if (opcode == DSP_DELAY_DP) take_data_odd(size * 2 + 1); else take_data(size + 1);
    put_offset(paramAddr, at);
}
void dsp_DELAY(int paramAddr)    { delay_param(paramAddr, DSP_DELAY); }
void dsp_DELAY_DP(int paramAddr) { delay_param(paramAddr, DSP_DELAY_DP); }

static int delay_word(unsigned short maxus, unsigned short us)
{
    need_plain_region();
    long long samples = ((long long)maxus * dspConvertFrequencyFromIndex(dspMaxSamplingFreq) + 500000) / 1000000;
    if (samples > 16000) fatal("delay too large.");
    return put_head((int)samples, us);                        /* [max samples : default microseconds] */
}
int dspDelay_MicroSec_Max(int maxus) { return delay_word((unsigned short)maxus, (unsigned short)maxus); }
int dspDelay_MicroSec_Max_Default(int maxus, int us) { return delay_word((unsigned short)maxus, (unsigned short)us); }
int dspDelay_MilliMeter_Max(int maxmm, float speed)
{
    return delay_word((unsigned short)(maxmm * 1000.0 / speed), (unsigned short)(maxmm * 1000.0 / speed));
}
int dspDelay_MilliMeter_Max_Default(int maxmm, int mm, float speed)
{
    return dspDelay_MicroSec_Max_Default((int)(maxmm * 1000.0 / speed), (int)(mm * 1000.0 / speed));
}

static unsigned delay_factor(int freq_index)                  /* 2^32 / 10^6 * fs, :1140-1146 */
{
    return (unsigned)(4294.967296 * dspConvertFrequencyFromIndex(freq_index));
}

static void delay_fixed(unsigned short microSec, int opcode)
{
    op_open(opcode);
    unsigned samples = (unsigned)(((unsigned long long)delay_factor(dspMaxSamplingFreq) * microSec) >> 32);
    addCode(microSec);
    if (opcode == DSP_DELAY) take_data(1 + (int)samples); else take_data_odd(1 + (int)samples * 2);
    addCode(0);                                               /* no parameter: the delay is fixed */
}
void dsp_DELAY_FixedMicroSec(int microSec)    { delay_fixed((unsigned short)microSec, DSP_DELAY); }
void dsp_DELAY_FixedMilliMeter(int mm, float speed) { dsp_DELAY_FixedMicroSec((int)(mm * 1000.0 / speed)); }
void dsp_DELAY_DP_FixedMicroSec(int microSec) { delay_fixed((unsigned short)microSec, DSP_DELAY_DP); }
void dsp_DELAY_DP_FixedMilliMeter(int mm, float speed) { dsp_DELAY_DP_FixedMicroSec((int)(mm * 1000.0 / speed)); }

void dsp_DATA_TABLE(int paramAddr, dspGainParam_t gain, int divider, int size)      /* :1178-1189 */
{
    int at = op_open(DSP_DATA_TABLE);
    if (paramAddr) find_region(paramAddr, size);
    put_param(gain);
    addCode(divider);
    addCode(size);
    take_data(1);
    put_offset(paramAddr, at);
    want_below_if0(paramAddr, size);
}

/* ---------------- biquads: :1212-1298 ---------------- */
int dsp_BIQUADS(int paramAddr)
{
    int at = op_open(DSP_BIQUADS);
    find_region_of(paramAddr, 2 + 6 * E.nf, DSP_BIQUADS);
    int num = E.w[paramAddr].s16.low;
    find_region(paramAddr, (2 + 6 * E.nf) * num);
    int state = take_data_even(num * 6);
    put_offset(paramAddr, at);
    return state + (num - 1) * 6;
}

int dspBiquad_Sections(int number)
{
    open_section(DSP_BIQUADS, number);
    int pos = pad_to_odd();
    E.sect.first = put_head(DSP_BIQUADS, number);
    addCode(1);                                               /* bypass flag: 1 = filter active */
    return pos;
}
int dspBiquad_Sections_Flexible(void) { return dspBiquad_Sections(0); }
int dspBiquad_Sections_Maximum(int number) { return dspBiquad_Sections(-number); }

static int put_filter_params(int type, dspFilterParam_t freq, dspFilterParam_t Q, dspGainParam_t gain)
{
    int at = put_head(type, (int)freq);
    if (!(at & 1)) fatal("Encoder bug (not expected). Adress should be misalligned here");
    addFloat((float)Q);
    addFloat(gain);
    return at;
}

static void put_biquad(dspFilterParam_t b0, dspFilterParam_t b1, dspFilterParam_t b2, dspFilterParam_t a1, dspFilterParam_t a2)
{
    pad_to_even();
    if (E.format < DSP_FORMAT_FLOAT) {
        addCode(DSP_QM32(b0, DSP_MANTBQ)); addCode(DSP_QM32(b1, DSP_MANTBQ)); addCode(DSP_QM32(b2, DSP_MANTBQ));
        addCode(DSP_QM32(a1 - 1.0, DSP_MANTBQ)); addCode(DSP_QM32(a2, DSP_MANTBQ));
    } else {
        addFloat((float)b0); addFloat((float)b1); addFloat((float)b2);
        addFloat((float)(a1 - 1.0)); addFloat((float)a2);
    }
}

/* ---------------- filter design: dsp_filters.c:18-175 ---------------- */
typedef struct { dspFilterParam_t b0, b1, b2, a1, a2; } biquad_t;

static biquad_t design_1st(int type, dspFilterParam_t fs, dspFilterParam_t freq, dspGainParam_t gain)
{
    biquad_t c = { 0, 0, 0, 0, 0 };
    dspFilterParam_t t = tan(M_PI * freq / fs), a0, k;
    switch (type) {
    case FLP1: k = 1.0 + t; c.a1 = (1.0 - t) / k; c.b0 = t / k * gain; c.b1 = c.b0; break;
    case FHP1: k = 1.0 + t; c.a1 = (1.0 - t) / k; c.b0 = 1.0 / k * gain; c.b1 = -1.0 / k * gain; break;
    case FHS1: { dspFilterParam_t A = sqrt(gain); a0 = A * t + 1.0;
                 c.a1 = -(A * t - 1.0) / a0; c.b0 = (A * t + gain) / a0; c.b1 = (A * t - gain) / a0; break; }
    case FLS1: { dspFilterParam_t A = sqrt(gain); a0 = t + A;
                 c.a1 = -(t - A) / a0; c.b0 = (gain * t + A) / a0; c.b1 = (gain * t - A) / a0; break; }
    case FAP1: k = (t - 1.0) / (t + 1.0); c.a1 = -k; c.b0 = k * gain; c.b1 = gain; break;
    default: break;
    }
    return c;
}

static biquad_t design_2nd(int type, dspFilterParam_t fs, dspFilterParam_t freq, dspFilterParam_t Q, dspGainParam_t gain)
{
    biquad_t c = { 0, 0, 0, 0, 0 };
    dspFilterParam_t w0 = M_PI * 2.0 * freq / fs, cw = cos(w0), sw = sin(w0);
    dspFilterParam_t alpha = (Q != 0.0) ? sw / 2.0 / Q : 1, a0 = 1.0 + alpha;
    c.a1 = -(-2.0 * cw) / a0;                                  /* feedback sign folded in, normalised by a0 */
    c.a2 = (alpha - 1.0) / a0;                                 /* written so that Q = 0 gives +0.0 like the reference's build */
    switch (type) {
    case FLP2:   c.b1 = (1.0 - cw) / a0 * gain;  c.b0 = c.b1 / 2.0;  c.b2 = c.b0; break;
    case FHP2:   c.b1 = -(1.0 + cw) / a0 * gain; c.b0 = -c.b1 / 2.0; c.b2 = c.b0; break;
    case FAP2:   c.b0 = -c.a2 * gain; c.b1 = -c.a1 * gain; c.b2 = gain; break;
    case FNOTCH: c.b0 = 1.0 / a0 * gain; c.b1 = -c.a1 * gain; c.b2 = c.b0; break;
    case FBPQ:   c.b0 = sw / 2.0 / a0; c.b1 = 0; c.b2 = -sw / 2.0 / a0; break;
    case FBP0DB: c.b0 = alpha / a0; c.b1 = 0; c.b2 = -alpha / a0; break;
    case FPEAK: {
        dspFilterParam_t A = sqrt(gain);
        a0 = 1.0 + alpha / A;
        c.a1 = 2.0 * cw / a0;
        c.a2 = -(1.0 - alpha / A) / a0;
        c.b0 = (1.0 + alpha * A) / a0; c.b1 = -2.0 * cw / a0; c.b2 = (1.0 - alpha * A) / a0;
        break; }
    case FLS2: {
        dspFilterParam_t A = sqrt(gain), r = sqrt(A);
        a0 = (A + 1.0) + (A - 1.0) * cw + 2.0 * r * alpha;
        c.a1 = -(-2.0 * ((A - 1.0) + (A + 1.0) * cw)) / a0;
        c.a2 = -((A + 1.0) + (A - 1.0) * cw - 2.0 * r * alpha) / a0;
        c.b0 = (A * ((A + 1.0) - (A - 1.0) * cw + 2.0 * r * alpha)) / a0;
        c.b1 = (2.0 * A * ((A - 1.0) - (A + 1.0) * cw)) / a0;
        c.b2 = (A * ((A + 1.0) - (A - 1.0) * cw - 2.0 * r * alpha)) / a0;
        break; }
    case FHS2: {
        dspFilterParam_t A = sqrt(gain), r = sqrt(A);
        a0 = (A + 1.0) - (A - 1.0) * cw + 2.0 * r * alpha;
        c.a1 = -(2.0 * ((A - 1.0) - (A + 1.0) * cw)) / a0;
        c.a2 = -((A + 1.0) - (A - 1.0) * cw - 2.0 * r * alpha) / a0;
        c.b0 = (A * ((A + 1.0) + (A - 1.0) * cw + 2.0 * r * alpha)) / a0;
        c.b1 = (-2.0 * A * ((A - 1.0) + (A + 1.0) * cw)) / a0;
        c.b2 = (A * ((A + 1.0) + (A - 1.0) * cw - 2.0 * r * alpha)) / a0;
        break; }
    default: break;
    }
    return c;
}

/* one biquad cell of the open bank: the descriptor once, then the coefficients for every encoded rate */
static int emit_cell(int order, int type, dspFilterParam_t freq, dspFilterParam_t Q, dspGainParam_t gain)
{
    int at = 0;
    section_entry(DSP_BIQUADS);
    for (int f = dspMinSamplingFreq; f <= dspMaxSamplingFreq; f++) {
        dspFilterParam_t fs = dspConvertFrequencyFromIndex(f);
        biquad_t c = order == 2 ? design_2nd(type, fs, freq, Q, gain) : design_1st(type, fs, freq, gain);
        if (!at) at = put_filter_params(type, freq, order == 2 ? Q : 0.0, gain);
        put_biquad(c.b0, c.b1, c.b2, c.a1, c.a2);
    }
    if (!E.sect.opcode) list_mark_here();
    return at;
}
int dsp_Filter2ndOrder(int type, dspFilterParam_t freq, dspFilterParam_t Q, dspGainParam_t gain) { return emit_cell(2, type, freq, Q, gain); }
int dsp_Filter1stOrder(int type, dspFilterParam_t freq, dspGainParam_t gain) { return emit_cell(1, type, freq, 0.0, gain); }

/* ---------------- presets: dsp_filters.c:244-507.  Each row: frequency factor, Q; a row with Q == 0 is the
 * first-order cell of the odd orders; `div` = the high-pass mirrors divide the frequency instead. ---------------- */
typedef struct { double k, q; } cell_t;

static int preset(int hp, dspFilterParam_t freq, const cell_t *cells, int n)
{
    int first = 0;
    for (int i = 0; i < n; i++) {
        dspFilterParam_t f = hp ? freq / cells[i].k : freq * cells[i].k;
        int at = cells[i].q != 0.0 ? dsp_Filter2ndOrder(hp ? FHP2 : FLP2, f, cells[i].q, 1.0)
                                   : dsp_Filter1stOrder(hp ? FHP1 : FLP1, f, 1.0);
        if (i == 0) first = at;
    }
    return first;
}
#define CELLS(...) (const cell_t[]){ __VA_ARGS__ }
#define N(...)     (int)(sizeof((const cell_t[]){ __VA_ARGS__ }) / sizeof(cell_t))
#define PRESET(name, hp, ...) int name(dspFilterParam_t freq) { return preset(hp, freq, CELLS(__VA_ARGS__), N(__VA_ARGS__)); }

#define BES2      { 1.0, 0.57735026919 }
#define BES3      { 0.941600026533, 0.691046625825 }, { 1.03054454544, 0.0 }
#define BES3_3DB  { 1.32267579991, 0.691046625825 }, { 1.44761713315, 0.0 }
#define BES4      { 0.944449808226, 0.521934581669 }, { 1.05881751607, 0.805538281842 }
#define BES4_3DB  { 1.43017155999, 0.521934581669 }, { 1.60335751622, 0.805538281842 }
#define BES6      { 0.928156550439, 0.510317824749 }, { 0.977488555538, 0.611194546878 }, { 1.10221694805, 1.02331395383 }
#define BES6_3DB  { 1.60391912877, 0.510317824749 }, { 1.68916826762, 0.611194546878 }, { 1.9047076123, 1.02331395383 }
#define BES8      { 0.920583104484, 0.505991069397 }, { 0.948341760923, 0.559609164796 }, { 1.01102810214, 0.710852074442 }, { 1.13294518316, 1.22566942541 }
#define BES8_3DB  { 1.77846591177, 0.505991069397 }, { 1.8320926012, 0.559609164796 }, { 1.95319575902, 0.710852074442 }, { 2.18872623053, 1.22566942541 }
#define BUT2      { 1.0, M_SQRT1_2 }
#define BUT3      { 1.0, 1.0 }, { 1.0, 0.0 }
#define BUT4      { 1.0, 0.54119610 }, { 1.0, 1.3065630 }
#define BUT6      { 1.0, 0.51763809 }, { 1.0, M_SQRT1_2 }, { 1.0, 1.9318517 }
#define BUT8      { 1.0, 0.50979558 }, { 1.0, 0.60134489 }, { 1.0, 0.89997622 }, { 1.0, 2.5629154 }
#define LR2       { 1.0, 0.5 }
#define LR3       { 1.0, 0.5 }, { 1.0, 0.0 }
#define LR4       { 1.0, M_SQRT1_2 }, { 1.0, M_SQRT1_2 }
#define LR6       { 1.0, 0.5 }, { 1.0, 1.0 }, { 1.0, 1.0 }

PRESET(dsp_LP_BES2, 0, BES2)          PRESET(dsp_HP_BES2, 1, BES2)
PRESET(dsp_LP_BUT2, 0, BUT2)          PRESET(dsp_HP_BUT2, 1, BUT2)
PRESET(dsp_LP_LR2, 0, LR2)            PRESET(dsp_HP_LR2, 1, LR2)
PRESET(dsp_LP_BES3, 0, BES3)          PRESET(dsp_HP_BES3, 1, BES3)
PRESET(dsp_LP_BES3_3DB, 0, BES3_3DB)  PRESET(dsp_HP_BES3_3DB, 1, BES3_3DB)
PRESET(dsp_LP_BUT3, 0, BUT3)          PRESET(dsp_HP_BUT3, 1, BUT3)
PRESET(dsp_LP_LR3, 0, LR3)            PRESET(dsp_HP_LR3, 1, LR3)
PRESET(dsp_LP_BES4, 0, BES4)          PRESET(dsp_HP_BES4, 1, BES4)
PRESET(dsp_LP_BES4_3DB, 0, BES4_3DB)  PRESET(dsp_HP_BES4_3DB, 1, BES4_3DB)
PRESET(dsp_LP_BUT4, 0, BUT4)          PRESET(dsp_HP_BUT4, 1, BUT4)
PRESET(dsp_LP_LR4, 0, LR4)            PRESET(dsp_HP_LR4, 1, LR4)
PRESET(dsp_LP_BES6, 0, BES6)          PRESET(dsp_HP_BES6, 1, BES6)
PRESET(dsp_LP_BES6_3DB, 0, BES6_3DB)  PRESET(dsp_HP_BES6_3DB, 1, BES6_3DB)
PRESET(dsp_LP_BUT6, 0, BUT6)          PRESET(dsp_HP_BUT6, 1, BUT6)
PRESET(dsp_LP_LR6, 0, LR6)            PRESET(dsp_HP_LR6, 1, LR6)
PRESET(dsp_LP_BES8, 0, BES8)          PRESET(dsp_HP_BES8, 1, BES8)
PRESET(dsp_HP_BES8_3DB, 1, BES8_3DB)
PRESET(dsp_LP_BUT8, 0, BUT8)          PRESET(dsp_HP_BUT8, 1, BUT8)

/* the -3 dB second-order Bessel pair scales the frequency first (dsp_filters.c:247-256) */
int dsp_LP_BES2_3DB(dspFilterParam_t freq) { return dsp_LP_BES2(freq * 1.27201964951); }
int dsp_HP_BES2_3DB(dspFilterParam_t freq) { return dsp_HP_BES2(freq / 1.27201964951); }
/* LR8 = two Butterworth-4 in series (:496-507) */
int dsp_LP_LR8(dspFilterParam_t freq) { int at = dsp_LP_BUT4(freq); dsp_LP_BUT4(freq); return at; }
int dsp_HP_LR8(dspFilterParam_t freq) { int at = dsp_HP_BUT4(freq); dsp_HP_BUT4(freq); return at; }

int dsp_filter(int type, dspFilterParam_t freq, dspFilterParam_t Q, dspGainParam_t gain)      /* :519-588 */
{
    switch (type) {
    case LPBE2: case LPBE3db2: return dsp_LP_BES2(freq);   case HPBE2: case HPBE3db2: return dsp_HP_BES2(freq);
    case LPBE3: case LPBE3db3: return dsp_LP_BES3(freq);   case HPBE3: case HPBE3db3: return dsp_HP_BES3(freq);
    case LPBE4: case LPBE3db4: return dsp_LP_BES4(freq);   case HPBE4: case HPBE3db4: return dsp_HP_BES4(freq);
    case LPBE6: case LPBE3db6: return dsp_LP_BES6(freq);   case HPBE6: case HPBE3db6: return dsp_HP_BES6(freq);
    case LPBE8: case LPBE3db8: return dsp_LP_BES8(freq);   case HPBE8: case HPBE3db8: return dsp_HP_BES8(freq);
    case LPBU2: return dsp_LP_BUT2(freq);   case HPBU2: return dsp_HP_BUT2(freq);
    case LPBU3: return dsp_LP_BUT3(freq);   case HPBU3: return dsp_HP_BUT3(freq);
    case LPBU4: return dsp_LP_BUT4(freq);   case HPBU4: return dsp_HP_BUT4(freq);
    case LPBU6: return dsp_LP_BUT6(freq);   case HPBU6: return dsp_HP_BUT6(freq);
    case LPBU8: return dsp_LP_BUT8(freq);   case HPBU8: return dsp_HP_BUT8(freq);
    case LPLR2: return dsp_LP_LR2(freq);    case HPLR2: return dsp_HP_LR2(freq);
    case LPLR3: return dsp_LP_LR3(freq);    case HPLR3: return dsp_HP_LR3(freq);
    case LPLR4: return dsp_LP_LR4(freq);    case HPLR4: return dsp_HP_LR4(freq);
    case LPLR6: return dsp_LP_LR6(freq);    case HPLR6: return dsp_HP_LR6(freq);
    case LPLR8: return dsp_LP_LR8(freq);    case HPLR8: return dsp_HP_LR8(freq);
    case FLP2: case FHP2: case FLS2: case FHS2: case FAP2: case FPEAK: case FNOTCH: case FBP0DB: case FBPQ:
        return dsp_Filter2ndOrder(type, freq, Q, gain);
    case FLP1: case FHP1: case FLS1: case FHS1: case FAP1:
        return dsp_Filter1stOrder(type, freq, gain);
    default:
        fprintf(stderr, "NOT SUPPORTED (type = %d)\n", type);
        return 0;
    }
}

/* ---------------- Hilbert pair: dsp_filters.c:218-240, design in encoder/dsp_HilbertDesign.c:1-155 ----------------
 * A 90-degree phase splitter made of two cascades of second-order all-pass cells H(z) = (c - z^-2) / (1 - c z^-2): the classic
 * polyphase half-band design of Valenzuela and Constantinides ("Digital signal processing schemes for efficient interpolation and
 * decimation", IEE Proc. G 130(6), 1983), whose 2N cell coefficients follow in closed form from the transition width through the
 * elliptic nome q (theta-function series for the Jacobi sn).  The odd-indexed coefficients are the reference path, the even ones
 * the +90 degree path; dsp_Hilbert(stages, transition, phase) emits ONE of the two cascades as a bank of `stages` cells, one
 * coefficient set per encoded sampling rate (transition is in Hz).
 *
 * Parity is byte identity with the reference encoder, so the arithmetic is the reference BINARY's, not a cleaner one: its design
 * file computes in single precision throughout (`#define double float`), calls the double libm (tan, sin, cos) on widened
 * arguments and narrows each result, and its -Ofast build (encoder/Makefile:18) takes both fourth roots as sqrt(sqrt(x)) in double,
 * multiplies by a float reciprocal 1/k where the source divides, and forms the series arguments as m * (pi / order).  Every
 * rounding below is where that build has one (checked against it for 2..10 stages, four encodings: tests/test_encoder.py).  The
 * series are summed until a term underflows to zero in single precision, as there (its 1e-100 bound is below every float). */
static float hilb_qpow(float q, long n)            /* q^n by binary powering, single precision */
{
    float z = 1.0f;
    while (n) { if (n & 1) z *= q; n >>= 1; q *= q; }
    return z;
}

static void hilbert_coefficients(float *coef, int ncoef, float transition)
{
    const float t2 = transition + transition;
    float k = (float)tan((double)(1.0f - t2) * (M_PI / 4));
    k *= k;
    const float kksqrt = (float)sqrt(sqrt((double)(1.0f - k * k)));
    const float e = (float)(((double)(1.0f - kksqrt) * 0.5) / (double)(kksqrt + 1.0f));
    const float e2 = e * e, e4 = e2 * e2;
    const float q = (((150.0f * e4 + 15.0f) * e4 + 2.0f) * e4 + 1.0f) * e;
    if (ncoef < 1) return;
    const int order = 2 * ncoef + 1;
    const double q4 = sqrt(sqrt((double)q));       /* q^(1/4) */
    const float inv_k = 1.0f / k;
    const double w = M_PI / (double)order;
    for (int c = 1; c <= ncoef; c++) {
        /* numerator series: sum (-1)^i q^(i(i+1)) sin((2i+1) c pi / order) */
        float num = 0.0f, term;
        int i = 0, sgn = 1;
        do {
            const float p = hilb_qpow(q, (long)i * (i + 1));
            term = (float)(((double)sgn * (double)p) * sin((double)((2 * i + 1) * c) * w));
            num += term;
            sgn = -sgn; i++;
        } while (fabs((double)term) > 1e-100);
        num = (float)((double)num * q4);
        /* denominator series: 1/2 + sum_{i>=1} (-1)^i q^(i^2) cos(2 i c pi / order) */
        float den = 0.0f;
        i = 1; sgn = -1;
        do {
            const float p = hilb_qpow(q, (long)i * i);
            term = (float)(((double)sgn * (double)p) * cos((double)(2 * i * c) * w));
            den += term;
            sgn = -sgn; i++;
        } while (fabs((double)term) > 1e-100);
        den += 0.5f;
        const float ww = num / den, wwsq = ww * ww;
        const float prod = (1.0f - wwsq * k) * (1.0f - wwsq * inv_k);
        const float x = (float)(sqrt((double)prod) / (double)(wwsq + 1.0f));
        coef[c - 1] = (1.0f - x) / (x + 1.0f);
    }
}

int dsp_Hilbert(int stages, dspFilterParam_t transition, dspGainParam_t phase)
{
    int at = 0;
    float coefs[20];                                           /* at most 10 stages, as in the reference */
    if (stages < 1 || stages > 10) fatal("dsp_Hilbert: 1 .. 10 stages.");
    for (int i = 0; i < stages; i++) {
        const int d = 2 * i + (phase == 0.0 ? 1 : 0);          /* odd coefficients: the reference path; even: +90 degrees */
        section_entry(DSP_BIQUADS);
        for (int f = dspMinSamplingFreq; f <= dspMaxSamplingFreq; f++) {
            dspFilterParam_t fs = dspConvertFrequencyFromIndex(f);
            hilbert_coefficients(coefs, stages * 2, (float)(transition / fs));
            if (f == dspMinSamplingFreq) at = put_filter_params(FHILB, 1000, transition, 1.0);
            /*          xn        xn-1  xn-2  yn-1  yn-2 */
            put_biquad(coefs[d],  0.0,  -1.0, 0.0,  coefs[d]);
        }
        if (!E.sect.opcode) list_mark_here();
    }
    return at;
}

/* ---------------- FIR: :1290-1373, with the impulse pointer fixed (see header) ---------------- */
int dspFir_Impulses(void)
{
    open_section(DSP_FIR, E.nf);
    int pos = pad_to_odd();
    E.sect.first = pos;
    put_head(DSP_FIR, E.nf);
    return pos;
}

int dspFir_Delay(int value)
{
    section_entry(DSP_FIR);
    int pos = pad_to_odd();
    if (value > 1) put_head(value, 0); else addCode(1);       /* [delay : 0] or a one-tap impulse */
    addCode(0);
    return pos;
}

int dspFir_ImpulseData(const float *taps, int length)
{
    section_entry(DSP_FIR);
    int pos = pad_to_odd();
    if (E.n + length >= E.cap) fatal("Fir impulse too large for the opcode table size.");
    addCode(length);
    for (int i = 0; i < length; i++) addFloat(taps[i]);
    return pos;
}

int dspFir_ImpulseFile(char *name, int length)
{
    FILE *f = fopen(name, "r");
    if (!f) fatal("cant open impulse file.");
    float *taps = (float *)malloc((size_t)(length > 0 ? length : 1) * sizeof(float));
    int got = 0;
    while (got < length && fscanf(f, "%f", &taps[got]) == 1) got++;
    fclose(f);
    if (got != length) fatal("Impulse file too small or access problem.");
    int pos = dspFir_ImpulseData(taps, length);
    free(taps);
    return pos;
}

void dsp_FIR(int paramAddr)
{
    int at = op_open(DSP_FIR);
    int span = find_region_of(paramAddr, 2 * E.nf, DSP_FIR);
    int end = span & 0xFFFF, longest = 0;
    int p = paramAddr + 1;                                    /* first impulse: the word behind the section header */
    for (int f = dspMinSamplingFreq; f <= dspMaxSamplingFreq; f++) {
        if (!(p & 1)) p++;                                    /* impulses start on odd words */
        if (p >= end) fatal("FIR Impulse list goes outside of PARAM section.");
        int length = E.w[p].s16.low, delay = E.w[p].s16.high, need;
        if (delay) { need = delay + 1; length = 1; } else need = length;
        if (need > longest) longest = need;
        put_offset(p, at);
        p += 1 + length;
    }
    take_data_even(longest);
}

/* ---------------- meters, DC blocker, dither, probes, generators: :1376-1547 ---------------- */
static void rms(int total_ms, int delay, int delay_is_steps, int sign)
{
    op_open(DSP_RMS);
    check_range(total_ms, 10, 7200000);
    double two32 = 4294967296.0, seconds = total_ms / 1000.0;
    if (!delay_is_steps) { check_range(delay, 1, total_ms); delay = total_ms / delay; }
    check_range(delay, 0, 1000);
    double steps = delay;
    take_data_odd(5 + 4 + delay * 2);                         /* counter, index, 3 sqrt words, sum, average, line */
    addCode(delay);
    for (int f = dspMinSamplingFreq; f <= dspMaxSamplingFreq; f++) {
        double fs = dspConvertFrequencyFromIndex(f);
        int count = (int)(delay ? fs * seconds / steps : fs * seconds);
        addCode(count);
        double n = count;
        float scale = delay ? (float)(two32 / sqrt(n * delay) + 0.5) : (float)(two32 / sqrt(n) + 0.5);
        int mult = (int)scale;
        addCode(mult * sign);
    }
    list_mark_here();
}
void dsp_RMS(int timems, int steps)        { rms(timems, steps, 1, 1); }
void dsp_RMS_MilliSec(int timems, int ms)  { rms(timems, ms, ms == 0, 1); }
void dsp_PWRXY(int timems, int steps)      { rms(timems, steps, 1, -1); }
void dsp_PWRXY_MilliSec(int timems, int ms){ rms(timems, ms, ms == 0, -1); }

void dsp_DCBLOCK(int lowfreq)
{
    op_open(DSP_DCBLOCK);
    check_range(lowfreq, 1, 100);
    float lowf = (float)lowfreq;
    take_data_even(4);
    for (int f = dspMinSamplingFreq; f <= dspMaxSamplingFreq; f++) {
        double fs = dspConvertFrequencyFromIndex(f);
        float pole = (float)(2.0 * M_PI * lowf / fs);
        put_param(-pole);
    }
}

void dsp_DITHER(void) { op_open(DSP_DITHER); take_data_even(6); }

void dsp_DITHER_NS2(int paramAddr)
{
    if (dspMinSamplingFreq < F44100 || dspMaxSamplingFreq > F192000)
        fatal("frequency range provided in encoderinit incompatible.");
    int at = op_open(DSP_DITHER_NS2);
    find_region(paramAddr, 3 * E.nf);
    take_data_even(3);
    put_offset(paramAddr, at);
}

void dsp_DISTRIB(int IO, int size)
{
    op_open(DSP_DISTRIB);
    check_io(IO);
    addCode(IO);
    mark_out(IO);
    check_range(size, 8, 1024);
    addCode(size);
    take_data(1 + size);
}

static void pulse(int opcode, int freq, dspGainParam_t gain)
{
    op_open(opcode);
    check_range(freq, 0, dspConvertFrequencyFromIndex(dspMinSamplingFreq) / 2);
    take_data(1);
    put_param(gain);
    for (int f = dspMinSamplingFreq; f <= dspMaxSamplingFreq; f++)
        addCode(dspConvertFrequencyFromIndex(f) / freq);
}
void dsp_DIRAC_Fixed(int freq, dspGainParam_t gain)      { pulse(DSP_DIRAC, freq, gain); }
void dsp_SQUAREWAVE_Fixed(int freq, dspGainParam_t gain) { pulse(DSP_SQUAREWAVE, freq, gain); }

void dsp_CLIP_Fixed(dspGainParam_t value)
{
    op_open(DSP_CLIP);
    if (value >= 1.0 || value <= -1.0) fatal("value not in range -0.999..+0.999.");
    put_param(value);
}

void dsp_SINE_Fixed(int freq, dspGainParam_t gain)
{
    op_open(DSP_SINE);
    check_range(freq, 20, dspConvertFrequencyFromIndex(dspMinSamplingFreq) / 4);
    take_data_even(4);
    put_param(gain);
    for (int f = dspMinSamplingFreq; f <= dspMaxSamplingFreq; f++) {
        float epsilon = (float)(2.0 * M_PI * (float)freq / (float)dspConvertFrequencyFromIndex(f));
        put_param(epsilon);
    }
}

/* ---------------- binary files: dsp_fileaccess.c:114-120, 147-158 ---------------- */
int dspCreateBuffer(char *name, int *buff, int size)
{
    FILE *f = fopen(name, "wb");
    if (!f) return -1;
    size_t n = fwrite(buff, sizeof(int), (size_t)size, f);
    fclose(f);
    return n == (size_t)size ? size : -1;
}

int dspReadBuffer(char *name, int *buff, int size)
{
    FILE *f = fopen(name, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (bytes > (long)size * (long)sizeof(int)) { fclose(f); return -1; }
    size_t n = fread(buff, 1, (size_t)bytes, f);
    fclose(f);
    return n == (size_t)bytes ? size : -1;
}
