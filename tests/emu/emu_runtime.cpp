/* tests/emu/emu_runtime.cpp -- fiber scheduler behind tests/emu/hip/hip_runtime.h.
 * TEST INFRASTRUCTURE ONLY. */
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <mutex>
#include <thread>
#include <vector>

namespace emu {

thread_local Idx threadIdx_, blockIdx_, blockDim_, gridDim_;

extern "C" void lbz_emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl lbz_emu_switch
.type lbz_emu_switch,@function
lbz_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size lbz_emu_switch,.-lbz_emu_switch
)");

enum { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
static const size_t STACK_BYTES = 128 * 1024;
static const unsigned MAX_THREADS = 1024;

struct Wave {
  unsigned long long slot[2][64];
  int pred[2][64];
  unsigned tag[2][64];    /* generation in which slot/pred was written (participation marker) */
  unsigned arrived;       /* lanes waiting at the current wave collective */
  unsigned live;          /* lanes not yet DONE */
  unsigned gen;           /* number of completed wave collectives */
  int site[64]; const char *sitefile[64];          /* LBZ_EMU_CHECK_SITES: where each lane entered the current collective */
};

struct Fiber {
  void *sp;
  int state;
  unsigned wave_gen_seen; /* wave generation this fiber waits to pass */
};

struct BlockCtx {
  Fiber fib[MAX_THREADS];
  Wave wave[MAX_THREADS / 64];
  char *stacks = nullptr;
  void *sched_sp;
  unsigned nthreads, cur;
  unsigned block_arrived, live;
  const std::function<void()> *body;
};

static thread_local BlockCtx *ctx;

static void fiber_main();
static void trampoline()
{
  fiber_main();
  abort();
}

static void yield_to_sched()
{
  BlockCtx *c = ctx;
  lbz_emu_switch(&c->fib[c->cur].sp, c->sched_sp);
}

static void fiber_main()
{
  BlockCtx *c = ctx;
  (*c->body)();
  unsigned t = c->cur;
  c->fib[t].state = DONE;
  c->live--;
  Wave &w = c->wave[t >> 6];
  w.live--;
  /* a finished lane may complete a pending wave collective / block barrier */
  if (w.live && w.arrived == w.live) { w.arrived = 0; w.gen++; }
  if (c->live && c->block_arrived == c->live) {
    c->block_arrived = 0;
    for (unsigned i = 0; i < c->nthreads; i++)
      if (c->fib[i].state == WAIT_BLOCK) c->fib[i].state = READY;
  }
  yield_to_sched();
}

void sync_block()
{
  BlockCtx *c = ctx;
  unsigned t = c->cur;
  c->block_arrived++;
  if (c->block_arrived == c->live) {
    c->block_arrived = 0;
    for (unsigned i = 0; i < c->nthreads; i++)
      if (c->fib[i].state == WAIT_BLOCK) c->fib[i].state = READY;
    return;
  }
  c->fib[t].state = WAIT_BLOCK;
  yield_to_sched();
}

/* all live lanes of the wave rendezvous; returns after everyone has arrived */
/* LBZ_EMU_CHECK_SITES=1: the lanes of a wave must enter a collective from the same place in the code -- lanes that come from
   different places have diverged (a collective under a lane-dependent condition), and what they exchange is not what a GPU
   wave would see */
static const int check_sites = getenv("LBZ_EMU_CHECK_SITES") ? atoi(getenv("LBZ_EMU_CHECK_SITES")) : 0;   /* 1: report each pair of places once, 2: abort */
static bool site_pair_is_new(const char *fa, int la, const char *fb, int lb)
{
  static std::mutex mu;
  static std::vector<std::pair<std::pair<const char *, int>, std::pair<const char *, int>>> seen;
  std::lock_guard<std::mutex> g(mu);
  for (auto &p : seen) if (p.first.first == fa && p.first.second == la && p.second.first == fb && p.second.second == lb) return false;
  seen.push_back({{fa, la}, {fb, lb}});
  return true;
}
thread_local int site_ = 0;
thread_local const char *sitefile_ = "";
static void wave_rendezvous_at(int site);
static void wave_rendezvous() { wave_rendezvous_at(site_); }
static void wave_rendezvous_at(int site)
{
  BlockCtx *c = ctx;
  unsigned t = c->cur;
  Wave &w = c->wave[t >> 6];
  unsigned my_gen = w.gen;
  w.site[t & 63] = site; w.sitefile[t & 63] = sitefile_;
  w.arrived++;
  if (w.arrived == w.live) {
    if (check_sites)
      for (unsigned l = 0; l < 64; l++) {
        const unsigned tt = (t & ~63u) + l;
        if (tt < c->nthreads && c->fib[tt].state != DONE && w.site[l] != site && site_pair_is_new(sitefile_, site, w.sitefile[l], w.site[l])) {
          fprintf(stderr, "emu: lanes %u and %u of block %u entered a wave collective from different places (%s:%d and %s:%d)\n",
                  t & 63, l, blockIdx_.x, sitefile_, site, w.sitefile[l], w.site[l]);
          if (check_sites > 1) abort();
          break;
        }
      }
    w.arrived = 0; w.gen++; return;
  }
  c->fib[t].state = WAIT_WAVE;
  c->fib[t].wave_gen_seen = my_gen;
  yield_to_sched();
}

void sync_wave() { wave_rendezvous(); }

unsigned long long wave_exchange(unsigned long long v, int src_lane)
{
  BlockCtx *c = ctx;
  unsigned t = c->cur, lane = t & 63;
  Wave &w = c->wave[t >> 6];
  unsigned my_gen = w.gen, buf = my_gen & 1;
  w.slot[buf][lane] = v;
  w.tag[buf][lane] = my_gen;
  wave_rendezvous();
  unsigned src = (unsigned)src_lane & 63;
  if (w.tag[buf][src] != my_gen) return v;          /* source lane did not take part */
  return w.slot[buf][src];
}

unsigned long long wave_ballot(int pred)
{
  BlockCtx *c = ctx;
  unsigned t = c->cur, lane = t & 63;
  Wave &w = c->wave[t >> 6];
  unsigned my_gen = w.gen, buf = my_gen & 1;
  w.pred[buf][lane] = pred != 0;
  w.tag[buf][lane] = my_gen;
  wave_rendezvous();
  unsigned long long m = 0;
  for (unsigned l = 0; l < 64; l++)
    if (w.tag[buf][l] == my_gen && w.pred[buf][l]) m |= 1ull << l;
  return m;
}

static void run_block(BlockCtx *c, const std::function<void()> &body, dim3 grid, dim3 block, unsigned bid)
{
  ctx = c;
  c->body = &body;
  c->nthreads = block.x;
  c->live = block.x;
  c->block_arrived = 0;
  blockIdx_ = {bid, 0, 0};
  blockDim_ = {block.x, 1, 1};
  gridDim_ = {grid.x, 1, 1};
  for (unsigned w = 0; w < (block.x + 63) / 64; w++) {
    unsigned n = block.x - w * 64;
    c->wave[w].arrived = 0;
    c->wave[w].gen = 1;
    memset(c->wave[w].tag, 0, sizeof c->wave[w].tag);
    c->wave[w].live = n < 64 ? n : 64;
  }
  for (unsigned t = 0; t < block.x; t++) {
    char *top = c->stacks + (size_t)(t + 1) * STACK_BYTES;
    void **sp = (void **)(((uintptr_t)top) & ~(uintptr_t)15);
    *--sp = nullptr;                    /* fake return address: keeps rsp = 8 mod 16 at entry */
    *--sp = (void *)trampoline;
    for (int i = 0; i < 6; i++) *--sp = nullptr;
    c->fib[t].sp = sp;
    c->fib[t].state = READY;
  }
  while (c->live) {
    bool progress = false;
    for (unsigned t = 0; t < c->nthreads; t++) {
      Fiber &f = c->fib[t];
      if (f.state == WAIT_WAVE && c->wave[t >> 6].gen != f.wave_gen_seen) f.state = READY;
      if (f.state != READY) continue;
      c->cur = t;
      threadIdx_ = {t, 0, 0};
      lbz_emu_switch(&c->sched_sp, f.sp);
      progress = true;
    }
    if (!progress) {
      fprintf(stderr, "emu: DEADLOCK in block %u (live=%u, at block barrier=%u)\n", bid, c->live, c->block_arrived);
      for (unsigned t = 0; t < c->nthreads; t += 64)
        fprintf(stderr, "  wave %u: live=%u arrived=%u state[lane0]=%d\n", t >> 6, c->wave[t >> 6].live,
                c->wave[t >> 6].arrived, c->fib[t].state);
      abort();
    }
  }
}

static BlockCtx *new_ctx()
{
  BlockCtx *c = new BlockCtx;
  c->stacks = (char *)mmap(nullptr, (size_t)MAX_THREADS * STACK_BYTES, PROT_READ | PROT_WRITE,
                           MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (c->stacks == MAP_FAILED) { perror("mmap"); abort(); }
  return c;
}

void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t)
{
  if (block.x > MAX_THREADS || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) {
    fprintf(stderr, "emu: unsupported launch geometry\n");
    abort();
  }
  unsigned nt = std::thread::hardware_concurrency();
  const char *env = getenv("LBZ_EMU_THREADS");
  if (env) nt = (unsigned)atoi(env);
  if (nt < 1) nt = 1;
  if (nt > grid.x) nt = grid.x;
  std::atomic<unsigned> next{0};
  /* block contexts (fiber stacks) are pooled across launches: a launch's worker threads are short-lived, the
     touched stack pages are not given back */
  static std::mutex pool_mu;
  static std::vector<BlockCtx *> pool;
  auto worker = [&]() {
    BlockCtx *my = nullptr;
    {
      std::lock_guard<std::mutex> lk(pool_mu);
      if (!pool.empty()) { my = pool.back(); pool.pop_back(); }
    }
    if (!my) my = new_ctx();
    for (;;) {
      unsigned b = next.fetch_add(1);
      if (b >= grid.x) break;
      run_block(my, body, grid, block, b);
    }
    std::lock_guard<std::mutex> lk(pool_mu);
    pool.push_back(my);
  };
  if (nt == 1) { worker(); return; }
  std::vector<std::thread> th;
  for (unsigned i = 0; i < nt; i++) th.emplace_back(worker);
  for (auto &t : th) t.join();
}

}  // namespace emu
