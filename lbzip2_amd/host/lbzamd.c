/*
 * lbzamd.c -- the command: lbzip2's option surface (reference src/main.c:322-627) and file handling (:635-905) over the
 * GPU batch path (lbzamd_io.c -> include/lbzip2_amd.h).  SURVEY.md 8 f-4.
 *
 * What is kept of the reference, because scripts and users depend on it:
 *   - the argument list is  $LBZIP2 $BZIP2 $BZIP argv[1..]  (tokens split at blanks and tabs, main.c:150-155, :337-354);
 *   - the program name selects the mode: bunzip2 / lbunzip2 decompress, bzcat / lbzcat decompress to stdout (:376-382);
 *   - -d -z -c -t -k -f -v -q -s -u -S -n N -m N -1..-9 -h -V -L and the long names --stdout --test --decompress --compress
 *     --fast --best --force --keep --small --sequential --verbose --quiet --help --version --license --repetitive-fast
 *     --repetitive-best --exponential; clusters of short options; "--" ends the options (:384-571);
 *   - no FILE: a filter; FILE operands: FILE -> FILE.bz2, FILE.bz2 -> FILE, .tbz .tbz2 .tz2 -> .tar, anything else ->
 *     .out (:635-683); operands with a compressed suffix are skipped when compressing; directories, links and files with
 *     several names are skipped unless -f / -k say otherwise (:699-760); the output is created exclusively with the
 *     owner's read/write bits and gets the input's owner, mode and times at the end (:781-872); the input is removed
 *     unless -k, -c or -t (:893-897); a partial output is removed when the program fails;
 *   - diagnostics are worded as lbzip2 words them ("skipping ...", "compressed data error: ...", the -v ratio line), exit
 *     status 0, 4 after a warning, 1 after a failure (main.c:54-55, signals.c:32).
 * What differs: -n and -m are parsed and range-checked as the reference does and then only bound the host's I/O threads
 * (the blocks are the device's); four long options of this program's own select devices and chunking (--devices=N,
 * --pipelines=N, --chunk-slabs=N, --report).  The reference's sources are not used: this file is written against its
 * behaviour, and tests/test_cli.py holds the two programs side by side (oracle/_ref/lbzip2_stock is the checker).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <signal.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/lbzip2_amd.h"
#include "lbzamd_io.h"

#define LBZAMD_VERSION "0.5"

enum { STATUS_OK = 0, STATUS_FAIL = 1, STATUS_WARN = 4 };
enum sink { TO_FILES, TO_STDOUT, TO_NOWHERE };

static struct {
  const char *prog;
  int decompress, force, keep, verbose, sequential, report;
  enum sink sink;
  unsigned level;
  uintmax_t threads, memory;         /* -n, -m: accepted as the reference accepts them */
  unsigned devices, pipelines, chunk_slabs;
  int warned;
} G = { .level = 9 };

static char *volatile g_partial;       /* the output file being written: removed if the program dies */

/* ------------------------------------------------------------------ diagnostics: "prog: [\"file\": ]text[: strerror]" */
static void say(const char *file, int quoted, int err, const char *fmt, va_list ap)
{
  fprintf(stderr, "%s: ", G.prog);
  if (file) fprintf(stderr, quoted ? "\"%s\": " : "%s: ", file);
  vfprintf(stderr, fmt, ap);
  if (err) fprintf(stderr, ": %s", strerror(err));
  fputc('\n', stderr);
  fflush(stderr);
}
static void drop_partial(void)
{
  char *p = g_partial;
  if (p) { g_partial = NULL; (void)unlink(p); }
}
static void die(const char *file, int quoted, int err, const char *fmt, ...) __attribute__((noreturn, format(printf, 4, 5)));
static void die(const char *file, int quoted, int err, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  if (err != EPIPE && err != EFBIG) say(file, quoted, err, fmt, ap);   /* a closed pipe is the reader's decision, not news */
  va_end(ap);
  drop_partial();
  _exit(STATUS_FAIL);
}
static void warn(int err, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void warn(int err, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  say(NULL, 0, err, fmt, ap);
  va_end(ap);
  G.warned = 1;
}
static void note(const char *file, int quoted, int err, const char *fmt, ...) __attribute__((format(printf, 4, 5)));
static void note(const char *file, int quoted, int err, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  say(file, quoted, err, fmt, ap);
  va_end(ap);
}

static void on_signal(int sig)
{
  char *p = g_partial;
  if (p) (void)unlink(p);
  signal(sig, SIG_DFL);
  raise(sig);
}

/* ------------------------------------------------------------------ the argument list */
struct words { const char **v; size_t n, cap; };
static void push(struct words *w, const char *s)
{
  if (w->n == w->cap) {
    w->cap = w->cap ? 2 * w->cap : 32;
    w->v = realloc(w->v, w->cap * sizeof *w->v);
    if (!w->v) die(NULL, 0, 0, "Insufficient memory to complete operation.");
  }
  w->v[w->n++] = s;
}

/* an integer with at most one binary suffix letter (k m g t p e, either case), within [lo, hi] */
static uintmax_t number(const char *text, char opt, uintmax_t lo, uintmax_t hi)
{
  static const char units[] = "KkMmGgTtPpEe";
  char *end = NULL;
  int ok = text[0] != '\0';
  uintmax_t v = 0;
  if (ok) {
    errno = 0;
    const long x = strtol(text, &end, 10);
    ok = errno == 0 && x >= 0 && (end[0] == '\0' || end[1] == '\0');
    v = (uintmax_t)x;
  }
  if (ok && end[0]) {
    const char *u = strchr(units, end[0]);
    ok = u != NULL;
    if (ok) {
      const unsigned shift = 10u * (unsigned)((u - units) / 2 + 1);
      ok = v <= (UINTMAX_MAX >> shift);
      v <<= shift;
    }
  }
  if (!ok || v < lo || v > hi)
    die(NULL, 0, 0, "failed to parse \"%s\" from \"-%c\" as an integer in [%ju..%ju], specify \"-h\" for help", text, opt, lo, hi);
  return v;
}

static void pick_sink(char which)                    /* -c / -t exclude each other; -t means decompress */
{
  if ((which == 'c' && G.sink == TO_NOWHERE) || (which == 't' && G.sink == TO_STDOUT))
    die(NULL, 0, 0, "\"-c\" and \"-t\" are incompatible, specify \"-h\" for help");
  if (which == 'c') G.sink = TO_STDOUT;
  else { G.sink = TO_NOWHERE; G.decompress = 1; }
}
static void pick_mode(int decompress)                /* -d / -z: an earlier -t no longer discards */
{
  G.decompress = decompress;
  if (G.sink == TO_NOWHERE) G.sink = TO_FILES;
}

static void print_help(void)
{
  printf(
    "Usage: %s [-n THREADS] [-k|-c|-t] [-d|-z] [-1 .. -9] [-f] [-u] [-v] [FILE ...]\n"
    "       %s -h | -V\n"
    "\n"
    "bzip2 compression on AMD Instinct GPUs; streams are byte-identical to lbzip2's.\n"
    "Invoked as bunzip2 / lbunzip2 it decompresses, as bzcat / lbzcat it decompresses to stdout.\n"
    "The words of $LBZIP2, $BZIP2 and $BZIP (separated by blanks or tabs) are read as arguments\n"
    "in front of the command line's.\n"
    "\n"
    "  -z, --compress      compress (the default unless the program name says otherwise)\n"
    "  -d, --decompress    decompress\n"
    "  -t, --test          decompress and check, write nothing; implies -k, excludes -c\n"
    "  -c, --stdout        write to standard output; implies -k, excludes -t\n"
    "  -k, --keep          keep the input files; also open files that have several names\n"
    "  -f, --force         open anything that can be opened, replace existing output files;\n"
    "                      with -dc copy input that is not bzip2 as it is\n"
    "  -1 .. -9            block size 100 kB .. 900 kB; --fast = -1, --best = -9 (default)\n"
    "  -u, --sequential    cut blocks where they are full instead of at every block size of\n"
    "                      input (bzip2's blocking; slightly smaller output)\n"
    "  -v, --verbose       say what is being done and how well it compressed\n"
    "  -n THREADS          bound on the host's I/O threads (the codec runs on the GPU)\n"
    "  -m SIZE, -s, --small, -S, -q, --quiet, --repetitive-fast, --repetitive-best,\n"
    "  --exponential       accepted as lbzip2 accepts them; no effect here\n"
    "  --devices=N         deal the work over N GPUs (0 = all); default: the current one\n"
    "  --pipelines=N       device contexts per GPU (default 2), --chunk-slabs=N slabs per call\n"
    "  --report            one line of timings on stderr per file\n"
    "  -h, --help          this text;  -V, -L, --version, --license   version and licence\n"
    "\n"
    "Without FILE: standard input to standard output.  FILE becomes FILE.bz2 and is removed;\n"
    "FILEs ending in .bz2 .tbz .tbz2 .tz2 are left alone.  Decompressing strips .bz2, turns\n"
    ".tbz .tbz2 .tz2 into .tar and appends .out to any other name.\n", G.prog, G.prog);
}
static void print_version(void)
{
  printf("lbzamd %s -- lbzip2-compatible bzip2 compressor for AMD Instinct (gfx950) GPUs\n"
         "An independent implementation of the bzip2 format; command line after lbzip2 2.5\n"
         "(https://github.com/kjn/lbzip2).  No warranty.\n", LBZAMD_VERSION);
}
static void finish_stdout_and_exit(void) __attribute__((noreturn));
static void finish_stdout_and_exit(void)
{
  if (fflush(stdout) || fclose(stdout)) die(NULL, 0, errno, "fclose(stdout)");
  _exit(STATUS_OK);
}

static int own_long_option(const char *name)
{
  const char *eq = strchr(name, '=');
  const size_t k = eq ? (size_t)(eq - name) : strlen(name);
  if (k == 6 && !strncmp(name, "report", k) && !eq) { G.report = 1; return 1; }
  if (!eq) return 0;
  if (k == 7 && !strncmp(name, "devices", k)) { G.devices = (unsigned)number(eq + 1, 'g', 0, 64) + 1000u; return 1; }   /* +1000: "given" */
  if (k == 9 && !strncmp(name, "pipelines", k)) { G.pipelines = (unsigned)number(eq + 1, 'p', 1, 64); return 1; }
  if (k == 11 && !strncmp(name, "chunk-slabs", k)) { G.chunk_slabs = (unsigned)number(eq + 1, 'c', 1, 4096); return 1; }
  return 0;
}

/* options out, operands left in `files` */
static void read_arguments(int argc, char **argv, struct words *files)
{
  static const char *const env_names[] = { "LBZIP2", "BZIP2", "BZIP" };
  struct words all = { 0 };
  for (size_t i = 0; i < sizeof env_names / sizeof env_names[0]; i++) {
    char *val = getenv(env_names[i]);
    if (!val) continue;
    for (char *tok = strtok(val, " \t"); tok; tok = strtok(NULL, " \t")) push(&all, tok);
  }
  for (int i = 1; i < argc; i++) push(&all, argv[i]);

  if (!strcmp(G.prog, "bunzip2") || !strcmp(G.prog, "lbunzip2")) G.decompress = 1;
  else if (!strcmp(G.prog, "bzcat") || !strcmp(G.prog, "lbzcat")) { G.decompress = 1; G.sink = TO_STDOUT; }

  /* sysconf(_SC_THREAD_THREADS_MAX) is "no limit" on Linux: the bound is what an unsigned holds (main.c:368-373) */
  uintmax_t most_threads = (uintmax_t)sysconf(_SC_THREAD_THREADS_MAX);
  if (most_threads > UINT_MAX) most_threads = UINT_MAX;

  int want_help = 0, want_version = 0;
  size_t i = 0;
  for (; i < all.n && !want_help && !want_version; i++) {
    const char *a = all.v[i];
    if (a[0] != '-') { push(files, a); continue; }
    if (a[1] == '-') {
      const char *name = a + 2;
      if (!*name) { i++; break; }                        /* "--": operands only from here */
      if (!strcmp(name, "stdout")) pick_sink('c');
      else if (!strcmp(name, "test")) pick_sink('t');
      else if (!strcmp(name, "decompress")) pick_mode(1);
      else if (!strcmp(name, "compress")) pick_mode(0);
      else if (!strcmp(name, "fast")) G.level = 1;
      else if (!strcmp(name, "best")) G.level = 9;
      else if (!strcmp(name, "force")) G.force = 1;
      else if (!strcmp(name, "keep")) G.keep = 1;
      else if (!strcmp(name, "small")) ;
      else if (!strcmp(name, "sequential")) G.sequential = 1;
      else if (!strcmp(name, "verbose")) G.verbose = 1;
      else if (!strcmp(name, "help")) want_help = 1;
      else if (!strcmp(name, "license") || !strcmp(name, "version")) want_version = 1;
      else if (!strcmp(name, "quiet") || !strcmp(name, "repetitive-fast") || !strcmp(name, "repetitive-best") || !strcmp(name, "exponential")) ;
      else if (own_long_option(name)) ;
      else die(NULL, 0, 0, "unknown option \"%s\", specify \"-h\" for help", a);
      continue;
    }
    for (const char *p = a + 1; *p; p++) {
      const char o = *p;
      if (o >= '1' && o <= '9') { G.level = (unsigned)(o - '0'); continue; }
      switch (o) {
      case 'c': case 't': pick_sink(o); continue;
      case 'd': pick_mode(1); continue;
      case 'z': pick_mode(0); continue;
      case 'f': G.force = 1; continue;
      case 'k': G.keep = 1; continue;
      case 'u': G.sequential = 1; continue;
      case 'v': G.verbose = 1; continue;
      case 's': case 'S': case 'q': continue;
      case 'h': want_help = 1; break;
      case 'L': case 'V': want_version = 1; break;
      case 'n': case 'm': {
        const char *val = p + 1;
        if (!*val) {                                     /* the value is the next word */
          if (i + 1 >= all.n) die(NULL, 0, 0, "option \"-%c\" requires an argument, specify \"-h\" for help", o);
          val = all.v[++i];
        }
        if (o == 'n') G.threads = number(val, o, 1, most_threads);
        else G.memory = number(val, o, 1, SIZE_MAX);
        break;
      }
      default:
        die(NULL, 0, 0, "unknown option \"-%c\", specify \"-h\" for help", o);
      }
      break;                                             /* -h, -V, -n, -m end their cluster */
    }
  }
  if (want_help) { print_help(); finish_stdout_and_exit(); }
  if (want_version) { print_version(); finish_stdout_and_exit(); }
  for (; i < all.n; i++) push(files, all.v[i]);
  free(all.v);

  if (G.sink == TO_FILES && files->n == 0) G.sink = TO_STDOUT;
  if (G.decompress) {
    if (files->n == 0 && isatty(STDIN_FILENO)) die(NULL, 0, 0, "won't read compressed data from a terminal, specify \"-h\" for help");
  } else if (G.sink == TO_STDOUT && isatty(STDOUT_FILENO))
    die(NULL, 0, 0, "won't write compressed data to a terminal, specify \"-h\" for help");
}

/* ------------------------------------------------------------------ names */
static const struct { const char *packed, *plain; int marks_compressed; } SUFFIXES[] = {
  { ".bz2", "", 1 }, { ".tbz2", ".tar", 1 }, { ".tbz", ".tar", 1 }, { ".tz2", ".tar", 1 }, { "", ".out", 0 },
};
static int ends_with(const char *s, const char *tail)
{
  const size_t a = strlen(s), b = strlen(tail);
  return a >= b && !strcmp(s + a - b, tail);
}
static int looks_compressed(const char *path)
{
  for (size_t i = 0; i < sizeof SUFFIXES / sizeof SUFFIXES[0]; i++)
    if (SUFFIXES[i].marks_compressed && ends_with(path, SUFFIXES[i].packed)) return 1;
  return 0;
}
static char *output_name(const char *path)
{
  char *out;
  if (!G.decompress) {
    if (asprintf(&out, "%s.bz2", path) < 0) die(NULL, 0, 0, "Insufficient memory to complete operation.");
    return out;
  }
  for (size_t i = 0;; i++)
    if (ends_with(path, SUFFIXES[i].packed)) {          /* the last entry matches everything */
      if (asprintf(&out, "%.*s%s", (int)(strlen(path) - strlen(SUFFIXES[i].packed)), path, SUFFIXES[i].plain) < 0)
        die(NULL, 0, 0, "Insufficient memory to complete operation.");
      return out;
    }
}

/* ------------------------------------------------------------------ one operand (NULL: the filter) */
static const char *const DATA_ERRORS[] = {          /* the reference's words for its enum error 3..19 (expand.c:69-94) */
  "bad stream header magic", "bad block header magic", "empty source alphabet", "bad number of trees", "no coding groups",
  "invalid selector", "invalid delta code", "invalid prefix code", "incomplete prefix code", "empty block", "unterminated block",
  "missing run length", "block CRC mismatch", "stream CRC mismatch", "block overflow", "primary index too large", "unexpected end of file",
};

static void transfer(int fd_in, const char *in_name, int in_quoted, int fd_out, const char *out_name, int out_quoted,
                     uintmax_t *bytes_in, uintmax_t *bytes_out)
{
  struct lbzamd_io_stats st;
  int sys = 0, code = 0, rc;
  char msg[256];
  if (G.verbose)
    note(NULL, 0, 0, "%s %s%s%s to %s%s%s", G.decompress ? "decompressing" : "compressing", in_quoted ? "\"" : "", in_name, in_quoted ? "\"" : "",
         out_quoted ? "\"" : "", out_name, out_quoted ? "\"" : "");
  if (G.decompress) {
    rc = lbzamd_io_decompress(fd_in, fd_out, G.force && fd_out == STDOUT_FILENO, G.report, &st, &sys, &code, msg, sizeof msg);
  } else {
    struct lbzamd_io_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.level = G.level;
    cfg.sequential = G.sequential;
    cfg.report = G.report;
    cfg.pipes = G.pipelines;
    cfg.chunk_slabs = G.chunk_slabs;
    if (G.devices >= 1000u) {
      const int have = lbzamd_device_count();
      const unsigned want = G.devices - 1000u;
      if (have < 1) die(NULL, 0, 0, "no HIP device (this program has no CPU path)");
      cfg.ndev = want == 0 || want > (unsigned)have ? (unsigned)have : want;
    }
    if (G.threads) { cfg.readers = G.threads < 16 ? (unsigned)G.threads : 16u; cfg.writers = G.threads < 4 ? (unsigned)G.threads : 4u; }
    rc = lbzamd_io_compress(fd_in, fd_out, &cfg, &st, &sys, msg, sizeof msg);
  }
  switch (rc) {
  case LBZAMD_IO_OK: break;
  case LBZAMD_IO_READ: die(in_name, in_quoted, sys, "read()");
  case LBZAMD_IO_WRITE: die(out_name, out_quoted, sys, "write()");
  case LBZAMD_IO_MEMORY: die(NULL, 0, 0, "Insufficient memory to complete operation. See manual page for ways of reducing memory usage.");
  case LBZAMD_IO_DATA:
    if (code == 3) die(in_name, in_quoted, 0, "not a valid bzip2 file");
    die(in_name, in_quoted, 0, "compressed data error: %s", code >= 3 && code <= 19 ? DATA_ERRORS[code - 3] : msg);
  default: die(NULL, 0, 0, "%s", msg[0] ? msg : "device error");
  }
  *bytes_in = st.in_bytes;
  *bytes_out = st.out_bytes;
}

static void one_operand(const char *path)
{
  struct stat sb;
  int fd_in = STDIN_FILENO, fd_out = -1;
  const char *in_name = "stdin", *out_name = "the bit bucket";
  int in_quoted = 0, out_quoted = 0;
  char *made = NULL;
  memset(&sb, 0, sizeof sb);

  if (path) {
    if (!G.force) {
      if (lstat(path, &sb)) { warn(errno, "skipping \"%s\": lstat()", path); return; }
      if (G.sink == TO_FILES && !S_ISREG(sb.st_mode)) { warn(0, "skipping \"%s\": not a regular file", path); return; }
      if (G.sink == TO_FILES && !G.keep && sb.st_nlink > 1) { warn(0, "skipping \"%s\": more than one links", path); return; }
    }
    if (!G.decompress && looks_compressed(path)) { warn(0, "skipping \"%s\": compressed suffix", path); return; }
    fd_in = open(path, O_RDONLY | O_NOCTTY);
    if (fd_in < 0) { warn(errno, "skipping \"%s\": open()", path); return; }
    if (fstat(fd_in, &sb)) {
      warn(errno, "skipping \"%s\": fstat()", path);
      if (close(fd_in)) die(NULL, 0, errno, "close(\"%s\")", path);
      return;
    }
    in_name = path;
    in_quoted = 1;
  }

  int have_output = 1;
  if (G.sink == TO_STDOUT) { fd_out = STDOUT_FILENO; out_name = "stdout"; }
  else if (G.sink == TO_FILES) {
    made = output_name(path);
    if (G.force && unlink(made) && errno != ENOENT) note(NULL, 0, errno, "unlink(\"%s\")", made);     /* explains the open() below, should it fail */
    fd_out = open(made, O_WRONLY | O_CREAT | O_EXCL, sb.st_mode & (S_IRUSR | S_IWUSR));
    if (fd_out < 0) {
      warn(errno, "skipping \"%s\": open(\"%s\")", path, made);
      free(made);
      made = NULL;
      have_output = 0;
    } else {
      g_partial = made;
      out_name = made;
      out_quoted = 1;
    }
  }

  if (have_output) {
    uintmax_t nin = 0, nout = 0;
    transfer(fd_in, in_name, in_quoted, fd_out, out_name, out_quoted, &nin, &nout);
    if (G.sink == TO_FILES) {
      /* the output takes the input's owner, permission bits and times (main.c:829-872) */
      if (fchown(fd_out, sb.st_uid, sb.st_gid)) warn(errno, "fchown(\"%s\")", made);
      else {
        if (sb.st_mode & (S_ISUID | S_ISGID | S_ISVTX)) warn(0, "\"%s\": won't restore any of setuid, setgid, sticky", made);
        if (fchmod(fd_out, sb.st_mode & (S_IRWXU | S_IRWXG | S_IRWXO))) warn(errno, "fchmod(\"%s\")", made);
      }
      const struct timespec ts[2] = { sb.st_atim, sb.st_mtim };
      if (futimens(fd_out, ts)) warn(errno, "futimens(\"%s\")", made);
      if (close(fd_out)) die(NULL, 0, errno, "close(\"%s\")", made);
      g_partial = NULL;
      free(made);
      if (!G.keep && unlink(path) && errno != ENOENT) warn(errno, "unlink(\"%s\")", path);
    }
    if (G.verbose && nin && nout) {
      const uintmax_t plain = G.decompress ? nout : nin, packed = G.decompress ? nin : nout;
      const double r = (double)packed / (double)plain;
      note(in_name, in_quoted, 0, "compression ratio is %s%.3f%s, space savings is %.2f%%", r < 1 ? "1:" : "", r < 1 ? 1 / r : r, r < 1 ? "" : ":1",
           100 * (1 - r));
    }
  }
  if (path && close(fd_in)) die(NULL, 0, errno, "close(\"%s\")", path);
}

int main(int argc, char **argv)
{
  const char *slash = strrchr(argv[0], '/');
  G.prog = slash ? slash + 1 : argv[0];
  signal(SIGPIPE, SIG_DFL);
  for (int s = 0; s < 4; s++) {
    static const int sigs[] = { SIGINT, SIGTERM, SIGHUP, SIGBUS };      /* (SIGBUS: a mapped input file that shrank under the program) */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_signal;
    sigaction(sigs[s], &sa, NULL);
  }
  struct words files = { 0 };
  read_arguments(argc, argv, &files);
  if (files.n == 0) one_operand(NULL);
  for (size_t i = 0; i < files.n; i++) one_operand(files.v[i]);
  if (G.sink == TO_STDOUT && close(STDOUT_FILENO)) die(NULL, 0, errno, "close(stdout)");
  _exit(G.warned ? STATUS_WARN : STATUS_OK);
}
