/*
 * pngloss_main.c -- `pngloss [options] -- pngfile [pngfile ...]` on an MI355X.
 *
 * Same command line, messages and exit codes as the reference tool (/root/reference/src/pngloss.c:28-165 usage and
 * argument checks, src/pngloss_opts.c:22-135 option table), but the per-file loop of pngloss_main_internal
 * (pngloss.c:168-223), which handles one file at a time, is replaced by a three-stage batch:
 *
 *     decode all inputs (libpng, worker threads)  ->  ONE batched call into libpngloss_hip.so  ->  encode (threads)
 *
 * The GPU call also returns the filtered scanlines of every image (colour type re-detected, per-row filters applied),
 * so the encode stage only deflates and frames chunks (png_stream_writer.c) -- libpng is used for decoding only.
 *
 * so that a GPU with 256 compute units works on up to 256 images at once (one workgroup per image).  Files are
 * processed in windows so that memory stays bounded.  Per-file semantics are unchanged: output naming (--ext / -o),
 * the overwrite rule, temp-file + atomic rename, --skip-if-larger, --strip, stdin/stdout with "-", and the exit code
 * is the error of the last failing file.  Verbose messages are buffered per file and printed in file order.
 */
#include <getopt.h>
#include <pthread.h>
#include "png_stream_reader.h"
#include <stdarg.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../../include/pngloss_hip.h"
#include "png_bridge.h"
#include "png_stream_writer.h"

#define PNGLOSS_VERSION "1.0.1-mi355x"
#define WINDOW_FILES window_files()      /* images per GPU batch (one workgroup each): 256, or $PNGLOSS_WINDOW_FILES */
static size_t window_files(void);

struct options {
    const char *extension, *output_path;
    char *const *files;
    unsigned long strength, bleed;
    unsigned num_files;
    bool from_stdin, to_stdout, force, skip_if_larger, strip, help, version, missing, verbose;
    bool gpu_deflate;          /* --gpu-deflate: IDAT data compressed on the device instead of by zlib level 9 */
    size_t total_files;        /* files of this call (how many GPUs are worth a context) */
    bool gpu_read;             /* --gpu-read: inverse filters + expansion to RGBA8 on the device (inflate stays zlib on the decode threads) */
};

static const char usage_text[] =
    "usage:  pngloss [options] -- pngfile [pngfile ...]\n"
    "        pngloss [options] - >stdout <stdin\n\n"
    "options:\n"
    "  -s, --strength 19 how much quality to sacrifice, from 0 to 100 (default 19)\n"
    "  -b, --bleed 2     bleed divider, from 1 (full dithering) to 32767 (none)\n"
    "  -f, --force       overwrite existing output files\n"
    "  -o, --output file destination file path to use instead of --ext\n"
    "  -v, --verbose     print status messages\n"
    "  -q, --quiet       don't print status messages (default, overrides -v)\n"
    "  -V, --version     print version number\n"
    "  --skip-if-larger  only save converted files if they're smaller than original\n"
    "  --ext new.png     set custom suffix/extension for output filenames\n"
    "  --strip           remove optional metadata (default on Mac)\n"
    "  --gpu-deflate     compress the image data on the GPU too (not zlib's bytes, same pixels,\n"
    "                    files several percent smaller than with zlib level 9, much faster)\n"
    "  --gpu-read        undo the PNG scanline filters and expand to RGBA on the GPU (plain,\n"
    "                    non-interlaced files; the others are read with libpng as usual)\n"
    "\n"
    "Lossily compresses PNGs by using more compressible colors that are close enough to the\n"
    "original values; the filter+quantise pass runs on the GPU (all files of a call as one batch).\n"
    "Output names end in \"-loss.png\" or your --ext; with \"-\" the image goes stdin -> stdout.\n"
    "Existing outputs are skipped unless --force is given.\n";

/* ------------------------------------------------------------------------------------------- options */

enum { OPT_EXT = 256, OPT_NO_FORCE, OPT_SKIP_LARGER, OPT_STRIP, OPT_GPU_DEFLATE, OPT_GPU_READ };

static bool parse_number(const char *text, unsigned long *out)
{
    char *end;
    unsigned long v = strtoul(text, &end, 10);
    if (end == text || *end) return false;
    *out = v;
    return true;
}

static pngloss_error parse_options(int argc, char **argv, struct options *o)
{
    static const struct option table[] = {
        { "strength", required_argument, NULL, 's' }, { "bleed", required_argument, NULL, 'b' },
        { "force", no_argument, NULL, 'f' },          { "no-force", no_argument, NULL, OPT_NO_FORCE },
        { "output", required_argument, NULL, 'o' },   { "ext", required_argument, NULL, OPT_EXT },
        { "verbose", no_argument, NULL, 'v' },        { "quiet", no_argument, NULL, 'q' },
        { "skip-if-larger", no_argument, NULL, OPT_SKIP_LARGER }, { "strip", no_argument, NULL, OPT_STRIP },
        { "version", no_argument, NULL, 'V' },        { "help", no_argument, NULL, 'h' },
        { "gpu-deflate", no_argument, NULL, OPT_GPU_DEFLATE },
        { "gpu-read", no_argument, NULL, OPT_GPU_READ },
        { NULL, 0, NULL, 0 },
    };
    for (int c; (c = getopt_long(argc, argv, "vqfo:Vhs:b:", table, NULL)) != -1;) {
        switch (c) {
        case 'v': o->verbose = true; break;
        case 'q': o->verbose = false; break;
        case 'f': o->force = true; break;
        case OPT_NO_FORCE: o->force = false; break;
        case OPT_EXT: o->extension = optarg; break;
        case OPT_SKIP_LARGER: o->skip_if_larger = true; break;
        case OPT_STRIP: o->strip = true; break;
        case OPT_GPU_DEFLATE: o->gpu_deflate = true; break;
        case OPT_GPU_READ: o->gpu_read = true; break;
        case 'h': o->help = true; break;
        case 'V': o->version = true; break;
        case 'o':
            if (o->output_path) { fputs("--output option can be used only once\n", stderr); return INVALID_ARGUMENT; }
            if (strcmp(optarg, "-") == 0) o->to_stdout = true;
            else o->output_path = optarg;
            break;
        case 's':
            if (!parse_number(optarg, &o->strength)) { fputs("-s, --strength requires a numeric argument\n", stderr); return INVALID_ARGUMENT; }
            break;
        case 'b':
            if (!parse_number(optarg, &o->bleed)) { fputs("-b, --bleed requires a numeric argument\n", stderr); return INVALID_ARGUMENT; }
            break;
        default: return INVALID_ARGUMENT;
        }
    }
    if (optind < argc) {
        int first = optind;
        if (first == argc - 1 && strcmp(argv[first], "-") == 0) {   /* lone "-": stdin -> stdout (or -> --output) */
            o->from_stdin = true;
            o->to_stdout = !o->output_path;
        }
        o->num_files = (unsigned)(argc - first);
        o->files = argv + first;
    } else if (optind <= 1) {
        o->missing = true;
    }
    return SUCCESS;
}

static void print_version_banner(FILE *f)
{
    fprintf(f, "pngloss, %s, filter+quantise path on AMD MI355X (%s).\n", PNGLOSS_VERSION, pngloss_hip_version());
    rwpng_version_info(f);
    fputs("\n", f);
}

/* ------------------------------------------------------------------------------------------- per-file job */

struct job {
    const char *in_name;      /* "stdin" for the pipe */
    char *out_name;           /* malloc'ed unless it aliases options.output_path */
    bool own_out_name;
    pngloss_error status;
    png24_image in, out;
    unsigned char *filters;
    unsigned char *line_types, *lines;   /* filtered scanlines from the GPU: type byte per row, width*4-pitched rows */
    size_t zsize;                        /* --gpu-deflate: `lines` holds the finished zlib stream of zsize bytes instead */
    int color_type;
    char *log;                /* buffered stderr text */
    size_t log_len;
    pngloss_hip_result gpu;
    png_stream_source src;    /* --gpu-read: inflated scanlines waiting for the device (src.scanlines != NULL) */
};

static void say(struct job *j, const char *fmt, ...)
{
    char line[1024];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    if (n <= 0) return;
    if ((size_t)n >= sizeof line) n = (int)sizeof line - 1;
    char *grown = realloc(j->log, j->log_len + (size_t)n + 1);
    if (!grown) return;
    memcpy(grown + j->log_len, line, (size_t)n + 1);
    j->log = grown;
    j->log_len += (size_t)n;
}

static void flush_log(struct job *j)
{
    if (j->log) fputs(j->log, stderr);
    free(j->log);
    j->log = NULL;
    j->log_len = 0;
}

static const char *leaf(const char *path)
{
    const char *s = strrchr(path, '/');
    return s ? s + 1 : path;
}

static char *with_extension(const char *name, const char *ext)
{
    size_t n = strlen(name);
    char *out = malloc(n + strlen(ext) + 1);
    if (!out) return NULL;
    memcpy(out, name, n + 1);
    if (n > 4 && (memcmp(out + n - 4, ".png", 4) == 0 || memcmp(out + n - 4, ".PNG", 4) == 0)) n -= 4;
    strcpy(out + n, ext);
    return out;
}

static bool exists(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (f) fclose(f);
    return f != NULL;
}

/* the private output copy of a decoded image (pngloss.c:471-484 of the reference) */
static void finish_decode(struct job *j, const struct options *o)
{
    if (j->status != SUCCESS) return;
    if (o->verbose) {
        say(j, "  read %luKB file\n", (unsigned long)((j->in.file_size + 500UL) / 1000UL));
        if (j->in.input_color == RWPNG_SRGB) say(j, "  passing sRGB tag from the input\n");
        else if (j->in.gamma != 0.45455) say(j, "  converted image from gamma %2.1f to gamma 2.2\n", 1.0 / j->in.gamma);
    }
    const size_t W = j->in.width, H = j->in.height;
    j->out.width = j->in.width;
    j->out.height = j->in.height;
    j->out.gamma = j->in.gamma;
    j->out.output_color = j->in.output_color;
    const size_t bytes = W * H * 4;
    j->out.rgba_data = malloc(bytes > 0 ? bytes : 1);
    j->out.row_pointers = malloc((H ? H : 1) * sizeof(unsigned char *));
    j->filters = malloc(H ? H : 1);
    j->line_types = malloc(H ? H : 1);
    j->lines = malloc(o->gpu_deflate ? pngloss_hip_zlib_bound((uint32_t)W, (uint32_t)H) : (bytes > 0 ? bytes : 1));
    if (!j->out.rgba_data || !j->out.row_pointers || !j->filters || !j->line_types || !j->lines) { j->status = OUT_OF_MEMORY_ERROR; return; }
    for (size_t y = 0; y < H; y++) {
        j->out.row_pointers[y] = j->out.rgba_data + y * W * 4;
        memcpy(j->out.row_pointers[y], j->in.row_pointers[y], W * 4);
    }
}

/* --gpu-read, host half: chunk walk + inflate (png_stream_reader.c); the image header fields the libpng path would have set
 * (rwpng.c:260-277: sRGB tag, else gAMA inside (0, 1], else the default) and the RGBA buffer the device will fill */
static bool decode_job_stream(struct job *j)
{
    if (!png_stream_read(j->in_name, &j->src)) return false;
    png24_image *im = &j->in;
    im->width = j->src.width; im->height = j->src.height; im->file_size = j->src.file_size;
    double gamma = 0.45455;
    if (j->src.has_srgb) im->input_color = im->output_color = RWPNG_SRGB;
    else {
        if (j->src.has_gama) gamma = j->src.gamma;
        if (gamma > 0 && gamma <= 1.0) im->input_color = im->output_color = RWPNG_GAMA_ONLY;
        else {
            say(j, "pngloss readpng:  ignored out-of-range gamma %f\n", gamma);
            im->input_color = im->output_color = RWPNG_NONE;
            gamma = 0.45455;
        }
    }
    im->gamma = gamma;
    const size_t W = im->width, H = im->height;
    im->rgba_data = malloc(W * H * 4);
    im->row_pointers = malloc(H * sizeof(unsigned char *));
    if (!im->rgba_data || !im->row_pointers) { free(j->src.scanlines); j->src.scanlines = NULL; j->status = OUT_OF_MEMORY_ERROR; return true; }
    for (size_t y = 0; y < H; y++) im->row_pointers[y] = im->rgba_data + y * W * 4;
    return true;
}

/* stage 1: open + decode + private output copy (pngloss.c:433-484 of the reference) */
static void decode_job(struct job *j, const struct options *o)
{
    if (j->status != SUCCESS) return;
    if (o->verbose) say(j, "%s:\n", j->in_name);
    if (o->gpu_read && !o->from_stdin && decode_job_stream(j)) return;       /* pixels follow from the device (decode_window_on_device) */
    FILE *f = o->from_stdin ? stdin : fopen(j->in_name, "rb");
    if (!f) {
        say(j, "  error: cannot open %s for reading\n", j->in_name);
        j->status = READ_ERROR;
        return;
    }
    pngloss_error rc = rwpng_read_image24(f, &j->in, o->strip, o->verbose);
    if (!o->from_stdin) fclose(f);
    if (rc != SUCCESS) {
        say(j, "  error: cannot decode image %s\n", o->from_stdin ? "from stdin" : leaf(j->in_name));
        j->status = rc;
        return;
    }
    finish_decode(j, o);
}

/* stage 3: encode to "<out>.tmp", rename over the destination (pngloss.c:379-431 of the reference) */
static pngloss_error encode_to(struct job *j, png24_image *img, unsigned char *filters, const struct options *o)
{
    FILE *f;
    char *tmp = NULL;
    if (o->to_stdout) {
        f = stdout;
        if (o->verbose) say(j, "  writing compressed image to stdout\n");
    } else {
        tmp = malloc(strlen(j->out_name) + 5);
        if (!tmp) return OUT_OF_MEMORY_ERROR;
        sprintf(tmp, "%s.tmp", j->out_name);
        f = fopen(tmp, "wb");
        if (!f) {
            say(j, "  error: cannot open '%s' for writing\n", tmp);
            free(tmp);
            return CANT_WRITE_ERROR;
        }
        if (o->verbose) say(j, "  writing compressed image as %s\n", leaf(j->out_name));
    }
    pngloss_error rc;
    if (filters) {
        /* the optimised image: scanlines were filtered on the GPU, only deflate + chunk framing happen here */
        const png_stream_image si = { img->width, img->height, j->color_type, j->line_types, j->lines, (size_t)img->width * 4, img->gamma,
                                      img->output_color != RWPNG_GAMA_ONLY && img->output_color != RWPNG_NONE, img->output_color == RWPNG_SRGB,
                                      img->chunks, img->maximum_file_size, o->gpu_deflate ? j->lines : NULL, o->gpu_deflate ? j->zsize : 0 };
        rc = png_stream_write(f, &si, &img->file_size, &img->metadata_size);
    } else {
        rc = rwpng_write_image24(f, img, NULL);     /* the untouched original (pipe fallback): plain libpng */
    }
    if (!o->to_stdout) {
        fclose(f);
        if (rc == SUCCESS && rename(tmp, j->out_name) != 0) rc = CANT_WRITE_ERROR;
        if (rc != SUCCESS) unlink(tmp);
    }
    free(tmp);
    if (rc != SUCCESS && rc != TOO_LARGE_FILE)
        say(j, "  error: failed writing image to %s (%d)\n", o->to_stdout ? "stdout" : j->out_name, (int)rc);
    return rc;
}

static void encode_job(struct job *j, const struct options *o)
{
    if (j->status != SUCCESS) return;
    if (o->verbose) say(j, "  compression complete\n  used %u unique symbols\n", j->gpu.unique_symbols);
    if (o->skip_if_larger) j->out.maximum_file_size = j->in.file_size - 1;
    j->out.chunks = j->in.chunks;          /* metadata travels to the output */
    j->in.chunks = NULL;
    pngloss_error rc = encode_to(j, &j->out, j->filters, o);
    if (o->verbose) {
        if (rc == SUCCESS) {
            say(j, "  wrote %luKB file (%.1f%% of original)\n", (unsigned long)((j->out.file_size + 500UL) / 1000UL),
                100.0f * (float)j->out.file_size / (float)j->in.file_size);
            if (j->out.metadata_size > 0) say(j, "  copied %dKB of additional PNG metadata\n", (int)(j->out.metadata_size + 500) / 1000);
        } else if (rc == TOO_LARGE_FILE) {
            say(j, "  file exceeded maximum size of %luKB\n", (unsigned long)((j->out.maximum_file_size + 500UL) / 1000UL));
        }
    }
    if (o->to_stdout && (rc == TOO_LARGE_FILE || rc == TOO_LOW_QUALITY)) {
        /* never leave a pipe empty: fall back to the decoded original */
        pngloss_error again = encode_to(j, &j->in, NULL, o);
        if (again != SUCCESS) rc = again;
    }
    j->status = rc;
}

/* ------------------------------------------------------------------------------------------- tiny parallel-for */

struct crew { struct job *jobs; size_t n, next; const struct options *o; void (*work)(struct job *, const struct options *); pthread_mutex_t mu; };

static void *crew_member(void *arg)
{
    struct crew *c = arg;
    for (;;) {
        pthread_mutex_lock(&c->mu);
        size_t i = c->next++;
        pthread_mutex_unlock(&c->mu);
        if (i >= c->n) return NULL;
        c->work(&c->jobs[i], c->o);
    }
}

static void for_each_job(struct job *jobs, size_t n, const struct options *o, void (*work)(struct job *, const struct options *))
{
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    size_t threads = cores > 1 ? (size_t)cores : 1;
    if (threads > n) threads = n;
    if (threads > 64) threads = 64;
    if (o->from_stdin || o->to_stdout) threads = 1;
    struct crew c = { jobs, n, 0, o, work, PTHREAD_MUTEX_INITIALIZER };
    if (threads <= 1) { crew_member(&c); return; }
    pthread_t tid[64];
    size_t started = 0;
    for (; started < threads; started++)
        if (pthread_create(&tid[started], NULL, crew_member, &c) != 0) break;
    if (!started) crew_member(&c);
    for (size_t t = 0; t < started; t++) pthread_join(tid[t], NULL);
}

/* ------------------------------------------------------------------------------------------- the batch driver */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static size_t window_files(void)
{
    const char *e = getenv("PNGLOSS_WINDOW_FILES");
    const long v = e ? atol(e) : 0;
    return v > 0 ? (size_t)v : 256;
}

/* stage 1 of a window, possibly running in the background while the previous window is on the GPU */
struct decode_ahead { struct job *jobs; size_t n; const struct options *o; pthread_t thread; bool running; double seconds; };

/* --gpu-read, device half: every file of the window whose scanlines were inflated, as ONE batch (its own context: the optimiser's
 * contexts may be busy with the previous window) */
static pngloss_hip_ctx *g_read_ctx = NULL;
static void decode_window_on_device(struct job *jobs, size_t n, const struct options *o)
{
    size_t m = 0;
    for (size_t i = 0; i < n; i++) if (jobs[i].src.scanlines && jobs[i].status == SUCCESS) m++;
    if (m) {
        pngloss_hip_png_source *src = calloc(m, sizeof *src);
        size_t *who = calloc(m, sizeof *who);
        int rc = src && who ? PNGLOSS_SUCCESS : PNGLOSS_OUT_OF_MEMORY_ERROR;
        if (rc == PNGLOSS_SUCCESS && !g_read_ctx) { g_read_ctx = pngloss_hip_create(-1); if (!g_read_ctx) rc = PNGLOSS_HIP_ERROR; }
        size_t k = 0;
        for (size_t i = 0; i < n && rc == PNGLOSS_SUCCESS; i++) {
            struct job *j = &jobs[i];
            if (!j->src.scanlines || j->status != SUCCESS) continue;
            src[k] = (pngloss_hip_png_source){ j->src.scanlines, j->src.width, j->src.height, j->src.color_type, j->src.bit_depth,
                                               j->src.palette_entries ? j->src.palette : NULL, j->src.palette_entries,
                                               j->src.has_trns ? j->src.trns : NULL, j->src.trns_bytes, j->in.rgba_data };
            who[k++] = i;
        }
        /* a status per file: a damaged file fails alone, like under the reference's one-file-at-a-time loop (pngloss.c:196-204) */
        int *st = calloc(m, sizeof *st);
        if (!st && rc == PNGLOSS_SUCCESS) rc = PNGLOSS_OUT_OF_MEMORY_ERROR;
        int brc = rc;
        if (rc == PNGLOSS_SUCCESS) brc = pngloss_hip_png_decode_batch_host_status(g_read_ctx, src, m, st);
        /* a failure of the batch as a whole arrives in every st[q] (pngloss_hip.h); a return code that no st[q] explains is treated the same
         * way -- never take a file for decoded on the strength of st[q] == 0 alone */
        int explained = 0;
        for (size_t q = 0; q < k && rc == PNGLOSS_SUCCESS; q++) if (st[q]) explained = 1;
        for (size_t q = 0; q < k; q++) {
            const int one = rc != PNGLOSS_SUCCESS ? rc : (st[q] ? st[q] : ((brc != PNGLOSS_SUCCESS && !explained) ? brc : PNGLOSS_SUCCESS));
            if (one != PNGLOSS_SUCCESS) { say(&jobs[who[q]], "  error: cannot decode image %s on the GPU (%d)\n", leaf(jobs[who[q]].in_name), one); jobs[who[q]].status = (pngloss_error)one; }
        }
        free(src); free(who); free(st);
    }
    for (size_t i = 0; i < n; i++) {
        if (!jobs[i].src.scanlines) continue;
        free(jobs[i].src.scanlines); jobs[i].src.scanlines = NULL;
        finish_decode(&jobs[i], o);
    }
}

static void *decode_ahead_main(void *arg)
{
    struct decode_ahead *d = arg;
    const double t0 = now_s();
    for_each_job(d->jobs, d->n, d->o, decode_job);
    if (d->o->gpu_read) decode_window_on_device(d->jobs, d->n, d->o);
    d->seconds = now_s() - t0;
    return NULL;
}

static void *warm_up_main(void *arg)
{
    (void)arg;
    if (pngloss_hip_device_count() > 0) {                    /* runtime, device context, code objects */
        pngloss_hip_ctx *c = pngloss_hip_create(-1);
        if (c) pngloss_hip_destroy(c);
    }
    return NULL;
}

static void decode_ahead_start(struct decode_ahead *d, struct job *jobs, size_t n, const struct options *o)
{
    d->jobs = jobs; d->n = n; d->o = o; d->seconds = 0;
    d->running = n && pthread_create(&d->thread, NULL, decode_ahead_main, d) == 0;
    if (n && !d->running) decode_ahead_main(d);              /* no thread: decode right here */
}

static void decode_ahead_wait(struct decode_ahead *d)
{
    if (d->running) pthread_join(d->thread, NULL);
    d->running = false;
}

/* stages 2 and 3 of a window whose files are decoded already */
static pngloss_error run_window(struct job *jobs, size_t n, const struct options *o, pngloss_hip_multi **ctx, double decode_seconds)
{
    const bool timing = getenv("PNGLOSS_TIMING") != NULL;
    const double t1 = now_s(), t0 = t1 - decode_seconds;

    /* stage 2: every decoded image of the window in one GPU batch */
    pngloss_hip_host_image *imgs = calloc(n ? n : 1, sizeof *imgs);
    pngloss_hip_scanlines *lines = calloc(n ? n : 1, sizeof *lines);
    pngloss_hip_zstream *zs = calloc(n ? n : 1, sizeof *zs);
    pngloss_hip_result *res = calloc(n ? n : 1, sizeof *res);
    size_t *who = calloc(n ? n : 1, sizeof *who), m = 0;
    if (!imgs || !lines || !zs || !res || !who) { free(imgs); free(lines); free(zs); free(res); free(who); return OUT_OF_MEMORY_ERROR; }
    for (size_t i = 0; i < n; i++)
        if (jobs[i].status == SUCCESS) {
            imgs[m] = (pngloss_hip_host_image){ jobs[i].out.rgba_data, jobs[i].filters, jobs[i].out.width, jobs[i].out.height };
            lines[m] = (pngloss_hip_scanlines){ jobs[i].line_types, jobs[i].lines, (size_t)jobs[i].out.width * 4, -1 };
            zs[m] = (pngloss_hip_zstream){ jobs[i].lines, pngloss_hip_zlib_bound(jobs[i].out.width, jobs[i].out.height), 0, -1, { 0, 0, 0 },
                                            PNGLOSS_HIP_Z_STREAM_ONLY };     /* the file is written from the stream alone */
            who[m++] = i;
        }
    if (m && o->gpu_deflate)
        for (size_t k = 0; k < m; k++)
            if (((uint64_t)imgs[k].width * 4 + 1) * imgs[k].height > ((uint64_t)1 << 30)) {
                /* the device encoder addresses one image's scanlines with 32-bit positions (include/pngloss_hip.h) */
                say(&jobs[who[k]], "  error: image too large for --gpu-deflate (more than 1 GiB of scanlines); run it without the option\n");
                jobs[who[k]].status = INVALID_ARGUMENT;
            }
    if (m && o->gpu_deflate) {                       /* drop the refused ones from the batch */
        size_t keep = 0;
        for (size_t k = 0; k < m; k++)
            if (jobs[who[k]].status == SUCCESS) { imgs[keep] = imgs[k]; lines[keep] = lines[k]; zs[keep] = zs[k]; who[keep++] = who[k]; }
        m = keep;
    }
    if (m) {
        /* every GPU of the node ($PNGLOSS_DEVICES restricts or repeats them): the files of the window are dealt out by size */
        const double tc0 = now_s();
        if (!*ctx) {
            /* a context (device initialisation, events, arenas) only on as many GPUs as the call has files for */
            const char *envd = getenv("PNGLOSS_DEVICES");
            const int ndev = pngloss_hip_device_count();
            if ((envd && *envd) || ndev <= 1 || (size_t)ndev <= o->total_files) *ctx = pngloss_hip_multi_create(NULL);
            else {
                char list[256]; size_t at = 0;
                for (size_t d = 0; d < o->total_files && at + 8 < sizeof list; d++) at += (size_t)snprintf(list + at, sizeof list - at, d ? ",%zu" : "%zu", d);
                *ctx = pngloss_hip_multi_create(list);
            }
        }
        if (timing) fprintf(stderr, "  [timing] GPU contexts ready after %.3f s\n", now_s() - tc0);
        int rc = !*ctx ? PNGLOSS_HIP_ERROR
               : pngloss_hip_multi_optimize_batch_host(*ctx, imgs, m, (unsigned)o->strength, (long)o->bleed, res,
                                                       o->gpu_deflate ? NULL : lines, o->gpu_deflate ? zs : NULL);
        for (size_t k = 0; k < m; k++) {
            jobs[who[k]].gpu = res[k];
            jobs[who[k]].color_type = o->gpu_deflate ? zs[k].color_type : lines[k].color_type;
            jobs[who[k]].zsize = zs[k].size;
            if (rc != PNGLOSS_SUCCESS && !(rc == PNGLOSS_INTERNAL_ABORT && res[k].status == 0)) {
                /* (a batch in which single images failed reports PNGLOSS_INTERNAL_ABORT and leaves the others done.)
                 * Unlike the reference (pngloss.c:266 ignores the return value) a failed optimisation is an error:
                 * there is no CPU path to fall back to, and writing an unoptimised file silently would be wrong */
                say(&jobs[who[k]], "  error: GPU optimisation failed (%d)\n", rc);
                jobs[who[k]].status = (pngloss_error)rc;
            }
        }
    }
    free(imgs); free(lines); free(zs); free(res); free(who);
    const double t2 = now_s();

    for_each_job(jobs, n, o, encode_job);
    if (timing) fprintf(stderr, "  [timing] %zu files: decode %.2f s, gpu batch %.2f s, encode %.2f s\n", n, t1 - t0, t2 - t1, now_s() - t2);
    return SUCCESS;
}

int main(int argc, char **argv)
{
    struct options o;
    memset(&o, 0, sizeof o);
    o.strength = 19;
    o.bleed = 2;
    pngloss_error rc = parse_options(argc, argv, &o);
    if (rc != SUCCESS) return rc;
    if (o.version) { puts(PNGLOSS_VERSION); return SUCCESS; }
    if (o.missing) { print_version_banner(stderr); fputs(usage_text, stderr); return MISSING_ARGUMENT; }
    if (o.help) { print_version_banner(stdout); fputs(usage_text, stdout); return SUCCESS; }
    if (o.strength > 255) { fputs("Must specify a strength in the range 0-255.\n", stderr); return INVALID_ARGUMENT; }
    if (o.bleed < 1 || o.bleed > 32767) { fputs("Must specify a bleed divider in the range 1-32767.\n", stderr); return INVALID_ARGUMENT; }
    if (o.extension && o.output_path) { fputs("--ext and --output options can't be used at the same time\n", stderr); return INVALID_ARGUMENT; }
    if (!o.extension) o.extension = "-loss.png";
    if (o.output_path && o.num_files != 1) {
        fputs("  error: Only one input file is allowed when --output is used. This error also happens when filenames with spaces are not in quotes.\n", stderr);
        return INVALID_ARGUMENT;
    }
    if (o.to_stdout && !o.from_stdin && o.num_files != 1) {
        fputs("  error: Only one input file is allowed when using the special output path \"-\" to write to stdout. This error also happens when filenames with spaces are not in quotes.\n", stderr);
        return INVALID_ARGUMENT;
    }
    if (!o.num_files && !o.from_stdin) {
        fputs("No input files specified.\n", stderr);
        if (o.verbose) print_version_banner(stderr);
        fputs(usage_text, stderr);
        return MISSING_ARGUMENT;
    }

    const size_t total = o.num_files;
    o.total_files = total;
    struct job *jobs = calloc(total ? total : 1, sizeof *jobs);
    if (!jobs) return OUT_OF_MEMORY_ERROR;
    for (size_t i = 0; i < total; i++) {
        struct job *j = &jobs[i];
        j->in_name = o.from_stdin ? "stdin" : o.files[i];
        if (!o.to_stdout) {
            if (o.output_path) j->out_name = (char *)o.output_path;
            else { j->out_name = with_extension(j->in_name, o.extension); j->own_out_name = true; }
            if (!j->out_name) j->status = OUT_OF_MEMORY_ERROR;
            else if (!o.force && exists(j->out_name)) {
                say(j, "  error: '%s' exists; not overwriting\n", j->out_name);
                j->status = NOT_OVERWRITING_ERROR;
            }
        }
    }

    pngloss_hip_multi *ctx = NULL;
    pngloss_error latest = SUCCESS;
    unsigned errors = 0, skipped = 0;
    /* windows of up to WINDOW_FILES files: decoded (threads), optimised as ONE GPU batch, encoded (threads).  The next
     * window is decoded in the background while the current one is on the GPU (pngloss.c:173 is a sequential loop). */
    struct decode_ahead ahead;
    memset(&ahead, 0, sizeof ahead);
    /* the HIP runtime needs a quarter of a second for its first call: let it be made while the first window is read and decoded */
    pthread_t warm;
    const bool warming = total && pthread_create(&warm, NULL, warm_up_main, NULL) == 0;
    decode_ahead_start(&ahead, jobs, total < WINDOW_FILES ? total : WINDOW_FILES, &o);
    for (size_t start = 0; start < total;) {
        size_t n = total - start < WINDOW_FILES ? total - start : WINDOW_FILES;
        decode_ahead_wait(&ahead);
        if (warming && start == 0) pthread_join(warm, NULL);
        const double decode_seconds = ahead.seconds;
        const size_t next = start + n, next_n = total - next < WINDOW_FILES ? total - next : WINDOW_FILES;
        if (next < total) decode_ahead_start(&ahead, jobs + next, next_n, &o);
        run_window(jobs + start, n, &o, &ctx, decode_seconds);
        for (size_t i = start; i < start + n; i++) {
            struct job *j = &jobs[i];
            flush_log(j);
            if (j->status != SUCCESS) {
                latest = j->status;
                if (j->status == TOO_LOW_QUALITY || j->status == TOO_LARGE_FILE) skipped++;
                else errors++;
            }
            rwpng_free_image24(&j->in);
            rwpng_free_image24(&j->out);
            free(j->filters);
            free(j->line_types);
            free(j->lines);
            if (j->own_out_name) free(j->out_name);
        }
        start += n;
    }
    if (ctx) pngloss_hip_multi_destroy(ctx);
    if (o.verbose) {
        const unsigned files = (unsigned)total;
        if (errors) fprintf(stderr, "There were errors compressing %d file%s out of a total of %d file%s.\n", errors, errors == 1 ? "" : "s", files, files == 1 ? "" : "s");
        if (skipped) fprintf(stderr, "Skipped %d file%s out of a total of %d file%s.\n", skipped, skipped == 1 ? "" : "s", files, files == 1 ? "" : "s");
        if (!skipped && !errors) fprintf(stderr, "Compressed %d image%s.\n", files, files == 1 ? "" : "s");
    }
    free(jobs);
    return latest;
}
