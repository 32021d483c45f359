/*
 * lbzamd_io.h -- the host splitter / muxer around the batch interface (include/lbzip2_amd.h), shared by the
 * `lbzamd` command (lbzamd.c) and the C test driver (lbzamd_compress.c).
 *
 * What lbzip2 does with one reader thread, N worker threads and one writer thread (src/process.c:260-307 source,
 * :351-417 sink, src/compress.c:238-250 in-order mux) is done here with
 *
 *   readers    a regular file is MAPPED: a chunk is a range of the mapping, nothing is read or page-locked (the runtime stages
 *              pageable memory across the link itself); LBZAMD_IO_NOMAP=1, or a file that cannot be mapped: R threads read
 *              it at chunk offsets with pread() -- any thread takes the next chunk -- into a ring of page-locked buffers.
 *              A pipe or a terminal is read by one thread in order into such a ring;
 *   pipelines  P per device, each with a device context of its own (created by the pipeline's thread, all at once):
 *              a chunk = a slab-aligned range compressed body-only (lbzamd_compress_host_body), so the H2D copies,
 *              kernels and D2H copies of consecutive chunks overlap;
 *   writers    W threads.  Where a chunk's bytes go in the stream is the sum of the sizes of the chunks in front of
 *              it, so a writer needs its predecessors' SIZES, not their bytes: on a regular file every finished chunk
 *              whose offset is known is written with pwrite(), in any order; a pipe is written by one thread in order.
 *              The stream CRC is folded from the 12-byte partials in chunk order (lbzamd_fold_parts; encode.h:38).
 *
 * The stream is the one the batch call, the work-unit interface and reference lbzip2 write for the same input.
 */
#ifndef LBZAMD_IO_H
#define LBZAMD_IO_H

#include <stddef.h>
#include <stdint.h>

struct lbzamd_io_cfg {
  unsigned level;          /* 1..9 */
  unsigned chunk_slabs;    /* slabs per chunk; 0 = 256, fewer for small regular files */
  unsigned pipes;          /* pipelines per device; 0 = 2 */
  unsigned ndev;           /* devices the pipelines are dealt over; 0 = the current device */
  unsigned readers;        /* 0 = 4 on a regular file; always 1 on a pipe */
  unsigned writers;        /* 0 = 2 on a regular file; always 1 on a pipe */
  int sequential;          /* lbzip2 -u: blocks cut where they are full -- the input as ONE batch call (a range of such a
                              stream cannot be cut ahead of time) */
  int report;              /* a timing line on stderr when done */
};

struct lbzamd_io_stats {
  uint64_t in_bytes, out_bytes;
  uint64_t chunks;
  unsigned readers, writers, pipelines, devices, chunk_slabs;
  double seconds;          /* wall clock, contexts and buffers included */
  double setup_seconds;    /* until the first pipeline had its context */
  double reader_busy;      /* seconds inside read()/pread(), summed over the readers */
  double writer_busy;      /* seconds inside write()/pwrite(), summed over the writers */
  double pipeline_busy;    /* seconds inside lbzamd_compress_host_body, summed over the pipelines */
};

enum { LBZAMD_IO_OK = 0, LBZAMD_IO_READ = 1, LBZAMD_IO_WRITE = 2, LBZAMD_IO_DEVICE = 3, LBZAMD_IO_MEMORY = 4,
       LBZAMD_IO_DATA = 5 /* decompress: the input is not what it should be; *err_code has the reference's enum error */ };

/* fd_in -> .bz2 stream on fd_out.  Returns LBZAMD_IO_*; on READ / WRITE *sys_errno is errno, on DEVICE the library's message
 * is lbzamd_last_error() of ... the thread that failed, copied to msg. */
int lbzamd_io_compress(int fd_in, int fd_out, const struct lbzamd_io_cfg *cfg, struct lbzamd_io_stats *st,
                       int *sys_errno, char *msg, size_t msg_cap);

/* .bz2 file(s) on fd_in -> bytes on fd_out (fd_out < 0: decode and check only, lbzip2 -t), in bounded memory: the input is
 * taken a window at a time (256 MB; LBZAMD_IO_DWINDOW=bytes), the whole blocks of a window are decoded at once
 * (lbzamd_decompress_window) and written before more is read -- a pipe of any length goes through, a reader that leaves
 * early (`| head`) ends the program.  An input of at most one window is decoded as one call decodes it.  not_bzip2_copy: a file that does not begin
 * with "BZh1".."BZh9" is copied through as it is instead of being refused (lbzip2 -dfc, process.c:675-678).
 * On LBZAMD_IO_DATA *err_code is the reference's enum error (3 = ERR_MAGIC: not a bzip2 file at all). */
int lbzamd_io_decompress(int fd_in, int fd_out, int not_bzip2_copy, int report, struct lbzamd_io_stats *st,
                         int *sys_errno, int *err_code, char *msg, size_t msg_cap);

#endif
