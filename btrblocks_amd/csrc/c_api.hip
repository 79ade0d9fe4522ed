// C ABI of the library (include/btrblocks_amd.h): host-side inspection, the CPU compressor
// entry points, and the GPU decode entry points. Decoding never falls back to the CPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/btrblocks_amd.h"
#include "btr_format.h"
#include "host/compress.h"
#include "plan.h"
#include "plan_builder.h"

using namespace btrfmt;

// The kernels live in the same translation unit so that no relocatable device code is needed.
#include "decode_kernels.hip"

namespace {

thread_local std::string g_error;

btr_status fail(btr_status st, const std::string& msg) {
  g_error = msg;
  return st;
}
btr_status hip_fail(hipError_t e, const char* what) {
  return fail(BTR_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return hip_fail(_e, #expr);  \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

btr_status parse_chunk(const uint8_t* p, size_t bytes, btr_chunk_info* out) {
  if (bytes < CHUNK_DATA) return fail(BTR_ERR_CORRUPT, "chunk shorter than its 12-byte envelope");
  out->scheme = p[CHUNK_SCHEME];
  out->nullmap_type = p[CHUNK_NM_TYPE];
  out->column_type = p[CHUNK_COL_TYPE];
  out->reserved = 0;
  out->tuple_count = rd32(p + CHUNK_TUPLES);
  if (out->column_type == COL_INTEGER)
    out->decoded_bytes = (uint64_t)out->tuple_count * 4;
  else if (out->column_type == COL_DOUBLE)
    out->decoded_bytes = (uint64_t)out->tuple_count * 8;
  else
    return fail(BTR_ERR_UNSUPPORTED, "only integer and double columns are supported");
  if (out->nullmap_type > NM_FLIPPED) return fail(BTR_ERR_CORRUPT, "unknown nullmap type");
  if ((size_t)rd32(p + CHUNK_NM_OFFSET) > bytes - CHUNK_DATA)
    return fail(BTR_ERR_CORRUPT, "nullmap offset past the end of the chunk");
  return BTR_OK;
}

ColumnSource chunk_source(const uint8_t* host, size_t bytes, uint64_t dev, const btr_chunk_info& info,
                          bool want_nullmap) {
  ColumnSource cs{};
  cs.host = host;
  cs.bytes = bytes;
  cs.dev_addr = dev;
  cs.payload_off = CHUNK_DATA;
  cs.nullmap_off = CHUNK_DATA + rd32(host + CHUNK_NM_OFFSET);
  cs.column_type = info.column_type;
  cs.scheme = info.scheme;
  cs.nullmap_type = info.nullmap_type;
  cs.tuple_count = info.tuple_count;
  cs.want_nullmap = want_nullmap;
  return cs;
}

btr_status parse_block_column(const uint8_t* p, size_t bytes, uint32_t column, btr_chunk_info* out,
                              uint32_t* payload_off, uint32_t* nullmap_off) {
  if (bytes < DB_ATTRS) return fail(BTR_ERR_CORRUPT, "block shorter than its header");
  const uint32_t n = rd32(p + DB_COUNT), cols = rd32(p + DB_COLUMNS);
  if (column >= cols) return fail(BTR_ERR_INVALID_ARGUMENT, "column index out of range");
  if ((size_t)cols * DB_ATTR_BYTES > bytes - DB_ATTRS)
    return fail(BTR_ERR_CORRUPT, "column table past the end of the block");
  const uint8_t* m = p + DB_ATTRS + (size_t)DB_ATTR_BYTES * column;
  out->scheme = m[CM_SCHEME];
  out->column_type = m[CM_COL_TYPE];
  out->nullmap_type = m[CM_NM_TYPE];
  out->reserved = 0;
  out->tuple_count = n;
  out->decoded_bytes = (uint64_t)n * (out->column_type == COL_DOUBLE ? 8 : 4);
  const uint32_t off = rd32(m + CM_OFFSET), nm = rd32(m + CM_NM_OFFSET);
  if (off > bytes || nm > bytes) return fail(BTR_ERR_CORRUPT, "column offset past the end of the block");
  if (out->nullmap_type > NM_FLIPPED) return fail(BTR_ERR_CORRUPT, "unknown nullmap type");
  if (payload_off) *payload_off = off;
  if (nullmap_off) *nullmap_off = nm;
  return BTR_OK;
}

btrhost::Options to_options(const btr_compress_options* o) {
  btrhost::Options opt;
  if (o) {
    if (o->max_cascade_depth) opt.max_cascade_depth = o->max_cascade_depth;
    opt.enable_for = o->enable_for != 0;
  }
  return opt;
}

}  // namespace

// ---- the allocation-free entry point's device memory ---------------------------------------------
// btr_gpu_decode_chunks_noalloc keeps everything a plan normally allocates (work items, launch
// order, status words, programs, table slices) in one file-scope device array. Every GPU holds
// its own copy at its own address: the address is taken once per device in btr_gpu_ctx_create
// (hipGetSymbolAddress with that device current) and host code only ever uses that address.
constexpr size_t kStaticBytes = 16u << 20;
constexpr size_t kStaticMaxChunks = 256;
constexpr size_t kStaticItemsOff = 0;                                          // WorkItem[256]
constexpr size_t kStaticOrderOff = kStaticMaxChunks * sizeof(WorkItem);        // LaunchEntry[256]
constexpr size_t kStaticStatusOff = kStaticOrderOff + kStaticMaxChunks * sizeof(LaunchEntry);  // u32[256]
constexpr size_t kStaticProgOff = 32u << 10;                                   // programs
constexpr size_t kStaticTablesOff = 1u << 20;                                  // table slices
static_assert(kStaticStatusOff + kStaticMaxChunks * 4 <= kStaticProgOff, "layout of the static area");
static __device__ __align__(256) unsigned char g_static_scratch[kStaticBytes];

namespace {
struct StaticArea {
  unsigned char* dev = nullptr;      // this device's copy of g_static_scratch
  unsigned char* staging = nullptr;  // pinned host mirror of [0, kStaticTablesOff)
  hipEvent_t uploaded = nullptr;     // the last call's upload has left the staging buffer
};
constexpr int kMaxDevices = 64;
StaticArea g_static_area[kMaxDevices];
std::mutex g_static_mutex;
}  // namespace

namespace {
// Kernel of a launch: a kernel kind, or one of the two mixed stream kernels.
constexpr uint32_t kMixedStream = KERNEL_KINDS, kMixedStreamNm = KERNEL_KINDS + 1;
// One kernel launch: the entries [begin, begin + count) of the launch order.
static_assert(kMaxSegments >= 5, "a mixed launch holds up to five stream kinds");
struct LaunchGroup {
  uint32_t kernel = 0;   // kernel kind, kMixedStream or kMixedStreamNm
  uint32_t threads = 0;  // workgroup size
  uint32_t lds = 0;      // dynamic LDS bytes: the largest need among its chunks
  size_t begin = 0, count = 0;
  uint64_t weight = 0;   // decoded bytes, scaled by how long a byte of this kind takes (launch_cost)
  LaunchSegments seg{};  // mixed kernels: which workgroups run which kind
};
}  // namespace

// Every stream the library uses besides the caller's lives in the context, created with it and
// destroyed with it. Plans and pipelines borrow them: nothing on the decode path creates or destroys
// a stream. All of them are plain non-blocking streams, which HIP multiplexes onto the process's
// hardware queues (four by default). They are never streams with a raised priority: such a stream
// gets a hardware queue of its own beyond those four, and a launch on the caller's stream that
// waits for an event of that queue was seen to never start once the other queues were in use (a
// plan of several kernel kinds decoded right behind other plans; the device then never went idle).
constexpr uint32_t kColumnLanes = 4;   // with the caller's stream: four launches beside each other
constexpr uint32_t kPipeStreams = 3;   // copy-in, decode and copy-out streams of the part pipelines

struct btr_gpu_ctx {
  int device = 0;
  StaticArea* area = nullptr;
  // Calls that enqueue on the context's streams do so under this mutex (the join events are reused
  // from call to call: a wait captures the record that precedes it on the host).
  std::recursive_mutex streams_mutex;
  hipStream_t lane[kColumnLanes] = {};
  hipEvent_t lane_done[kColumnLanes] = {};
  hipEvent_t lanes_fork = nullptr;
  hipStream_t pipe[kPipeStreams] = {};
  uint32_t next_side = 0;
  ~btr_gpu_ctx() {
    DeviceGuard guard(device);
    auto drop = [](hipStream_t& s) {
      if (!s) return;
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
      s = nullptr;
    };
    for (hipStream_t& s : lane) drop(s);
    for (hipStream_t& s : pipe) drop(s);
    for (hipEvent_t& e : lane_done)
      if (e) (void)hipEventDestroy(e);
    if (lanes_fork) (void)hipEventDestroy(lanes_fork);
  }
};

struct btr_gpu_plan {
  int device = 0;
  size_t n_items = 0;
  WorkItem* d_items = nullptr;
  uint8_t* d_programs = nullptr;
  size_t program_bytes = 0;
  uint32_t* d_status = nullptr;
  LaunchEntry* d_order = nullptr;  // launch entries (work item index, program place), launch by launch
  uint8_t* d_scratch = nullptr;
  uint32_t* d_debug = nullptr;  // BTR_DEBUG_FLAGS & 8: kDebugSlots timestamps per work item
  // The launches of a decode, the one with the most decoded bytes first. A plan with several of
  // them runs them beside each other on lanes of the context, which fork from and join the
  // caller's stream by the plan's events.
  std::vector<LaunchGroup> launches;
  std::vector<uint32_t> lane;       // per launch: 0 = the caller's stream, k = lane k - 1 of the context
  std::vector<hipEvent_t> done;     // per launch that runs on a lane: recorded behind it
  std::vector<hipEvent_t> go;       // small side launches: recorded right before the launch; the main launch waits for it
  hipEvent_t fork = nullptr;
  bool balanced = false;  // several large launches (a relation-level plan): no small odd launches to favour
  btr_plan_stats stats{};
  std::vector<uint32_t> h_status;
};

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Header walk and validation of every chunk; chunks are independent, so large plans are compiled
// on several host threads (the first failing chunk is the one reported). Nothing thrown inside
// escapes: a hostile header that exhausts memory is reported like any other malformed one.
btr_status compile_chunks(const btr_chunk_desc* chunks, size_t n, std::vector<WorkItem>& items,
                          std::vector<BuiltProgram>& progs, btr_plan_stats& stats, size_t max_workers = 16) {
  items.assign(n, WorkItem{});
  progs.assign(n, BuiltProgram{});
  std::vector<btr_chunk_info> infos(n);
  auto compile = [&](size_t i, std::string& err) -> btr_status {
    const btr_chunk_desc& d = chunks[i];
    if (!d.host_src || !d.dev_src || !d.dev_dst) {
      err = "null pointer";
      return BTR_ERR_INVALID_ARGUMENT;
    }
    btr_chunk_info& info = infos[i];
    btr_status st = parse_chunk((const uint8_t*)d.host_src, d.src_bytes, &info);
    if (st != BTR_OK) {
      err = g_error;
      return st;
    }
    if (d.dst_bytes < info.decoded_bytes) {
      err = "destination too small";
      return BTR_ERR_BUFFER_TOO_SMALL;
    }
    if ((uint64_t)(uintptr_t)d.dev_dst % 8 != 0) {
      err = "destination must be 8-byte aligned";
      return BTR_ERR_MISALIGNED;
    }
    if (d.src_bytes > 0xFFFFFFFFull) {
      err = "chunk above 4 GiB";
      return BTR_ERR_UNSUPPORTED;
    }
    try {
      build_program(chunk_source((const uint8_t*)d.host_src, d.src_bytes, (uint64_t)(uintptr_t)d.dev_src, info,
                                 d.dev_nullmap_dst != nullptr),
                    progs[i]);
    } catch (const PlanError& e) {
      err = e.what();
      return (btr_status)e.status;
    } catch (const std::bad_alloc&) {
      err = "out of host memory while compiling the decode program";
      return BTR_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {
      err = e.what();
      return BTR_ERR_CORRUPT;
    } catch (...) {
      err = "unknown failure while compiling the decode program";
      return BTR_ERR_CORRUPT;
    }
    WorkItem& it = items[i];
    it.src = (uint64_t)(uintptr_t)d.dev_src;
    it.dst = (uint64_t)(uintptr_t)d.dev_dst;
    it.nullmap_dst = (uint64_t)(uintptr_t)d.dev_nullmap_dst;
    it.tuple_count = info.tuple_count;
    it.src_bytes = (uint32_t)d.src_bytes;
    return BTR_OK;
  };
  const size_t hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t workers = n < 256 || max_workers <= 1 ? 1 : std::min<size_t>({hw, max_workers, n / 128});
  std::vector<size_t> bad_at(workers, n);
  std::vector<btr_status> bad_status(workers, BTR_OK);
  std::vector<std::string> bad_text(workers);
  auto run = [&](size_t w) {
    const size_t lo = n * w / workers, hi = n * (w + 1) / workers;
    for (size_t i = lo; i < hi; i++) {
      const btr_status st = compile(i, bad_text[w]);
      if (st != BTR_OK) {
        bad_at[w] = i;
        bad_status[w] = st;
        return;
      }
    }
  };
  if (workers == 1) {
    run(0);
  } else {
    std::vector<std::thread> pool;
    for (size_t w = 0; w < workers; w++) pool.emplace_back(run, w);
    for (auto& t : pool) t.join();
  }
  for (size_t w = 0; w < workers; w++)  // ranges are in chunk order: the first failure is the lowest index
    if (bad_status[w] != BTR_OK) return fail(bad_status[w], "chunk " + std::to_string(bad_at[w]) + ": " + bad_text[w]);
  for (size_t i = 0; i < n; i++) {
    stats.tuples += infos[i].tuple_count;
    stats.decoded_bytes += infos[i].decoded_bytes;
    stats.stored_bytes += chunks[i].src_bytes;
    stats.needed_bytes += CHUNK_DATA;
    if (chunks[i].dev_nullmap_dst) stats.nullmap_bytes += infos[i].tuple_count;
  }
  return BTR_OK;
}

uint32_t debug_flags_from_env() {
  const char* dbg = getenv("BTR_DEBUG_FLAGS");  // kernel timing diagnostics, see plan.h
  uint32_t flags = dbg ? (uint32_t)atoi(dbg) : 0;
#ifndef BTR_DEBUG_STEPS
  flags &= 8;  // step skipping exists in diagnostic builds only
#endif
  return flags;
}

// Program blob, launch order and launches of a set of compiled chunks.
struct Layout {
  std::vector<uint8_t> blob;
  std::vector<LaunchEntry> order;
  std::vector<LaunchGroup> launches;  // the one with the most estimated work (LaunchGroup::weight) first
  uint64_t scratch = 0;
};

unsigned threads_of(uint32_t kind) {
  return kind == KERNEL_STREAM_DICTD || kind == KERNEL_STREAM_DICTD + KERNEL_STREAM_NULLMAP ? kDictdThreads
         : kernel_is_generic(kind)                                                         ? kBlockThreads
                                                                                           : kStreamThreads;
}
bool kind_is_double(uint32_t kind) {
  const uint32_t k = kind >= KERNEL_STREAM_NONE + KERNEL_STREAM_NULLMAP ? kind - KERNEL_STREAM_NULLMAP : kind;
  return k == KERNEL_GENERIC_DBL || k == KERNEL_STREAM_PSEUDO || k == KERNEL_STREAM_DICTD || k == KERNEL_PFOR_DBL;
}
// Workgroups of a kernel kind that share a CU when each needs `lds` bytes of LDS. The dynamic LDS
// size belongs to the launch, so chunks that need little would be held to the occupancy of the
// neediest chunk of their launch: chunks are grouped into launches by this figure.
uint32_t workgroups_per_cu(uint32_t kind, uint32_t lds) {
  const uint32_t by_waves = kind == KERNEL_GENERIC_INT ? 6 : kernel_is_generic(kind) ? 4 : threads_of(kind) == kDictdThreads ? 2 : 4;
  const uint32_t by_lds = kLdsCapBytes / std::max<uint32_t>(512, (lds + 511) / 512 * 512);
  return std::max<uint32_t>(1, std::min(by_waves, by_lds));
}
// Rough relative time to decode one byte with a kernel kind (measured on whole columns; only the
// ratios matter): used to deal launches to lanes so that the lanes finish together.
uint32_t launch_cost(uint32_t kind, uint32_t per_cu) {
  switch (kind >= KERNEL_STREAM_NONE + KERNEL_STREAM_NULLMAP ? kind - KERNEL_STREAM_NULLMAP : kind) {
    case KERNEL_GENERIC_INT: return 22;
    case KERNEL_GENERIC_DBL: return per_cu <= 3 ? 40 : 22;  // programs with many tables: nested Pseudodecimal
    case KERNEL_STREAM_PSEUDO: return 34;
    case KERNEL_STREAM_DICTD: return 25;
    case KERNEL_PFOR_INT:
    case KERNEL_PFOR_DBL: return 33;
    default: return 29;
  }
}
// The 512-thread stream kinds can share a launch (the mixed kernels); 0 = the kind launches alone.
uint32_t mixed_kernel_of(uint32_t kind) {
  if (kind >= KERNEL_STREAM_NONE && kind <= KERNEL_STREAM_PSEUDO) return kMixedStream;
  if (kind >= KERNEL_STREAM_NONE + KERNEL_STREAM_NULLMAP && kind <= KERNEL_STREAM_PSEUDO + KERNEL_STREAM_NULLMAP) return kMixedStreamNm;
  return 0;
}

void lay_out(std::vector<WorkItem>& items, const std::vector<BuiltProgram>& progs, btr_plan_stats& stats, Layout& lo) {
  size_t total = 0;
  for (const BuiltProgram& p : progs) total += (p.program.size() + 15) / 16 * 16;
  lo.blob.assign(total, 0);
  size_t at = 0;
  for (size_t i = 0; i < items.size(); i++) {
    items[i].flags = 0;
    items[i].prog_off = (uint32_t)at;
    items[i].prog_bytes = (uint32_t)progs[i].program.size();
    memcpy(lo.blob.data() + at, progs[i].program.data(), progs[i].program.size());
    at += (progs[i].program.size() + 15) / 16 * 16;
    items[i].scratch = lo.scratch;  // offset for now; rebased by the caller
    lo.scratch += ((uint64_t)progs[i].scratch_bytes + 255) / 256 * 256;
    stats.max_lds_bytes = std::max(stats.max_lds_bytes, progs[i].lds_bytes);
    stats.max_depth = std::max(stats.max_depth, progs[i].depth);
    stats.needed_bytes += progs[i].needed_bytes;
    stats.kernel_mask |= 1u << progs[i].kernel;
  }
  stats.scratch_bytes = lo.scratch;
  stats.chunks = items.size();
  // A launch holds the chunks of one kernel (a kernel kind, or the stream kinds a mixed kernel
  // covers) that run the same number of workgroups per CU. Inside a launch the kinds whose blocks
  // take longest come first (the workgroups handed out last are then the short ones), chunks of a
  // kind in the order they were given: neighbouring workgroups then write neighbouring blocks of a
  // column (sorting a column's blocks by their table sizes instead cost its launch 8-17 %).
  struct Key {
    uint32_t kernel, per_cu;
    bool operator==(const Key& o) const { return kernel == o.kernel && per_cu == o.per_cu; }
  };
  std::vector<Key> keys;
  std::vector<std::vector<uint32_t>> members;
  // Chunks that all take one kernel stay one launch whatever their LDS needs (the blocks of one
  // column, some of which cross an occupancy step: two launches would have to fork and join, which
  // costs more than the few workgroups per CU gained).
  bool one_kernel = true;
  for (size_t i = 1; i < items.size() && one_kernel; i++) {
    const uint32_t a = progs[i].kernel, b = progs[0].kernel;
    one_kernel = (mixed_kernel_of(a) ? mixed_kernel_of(a) : a) == (mixed_kernel_of(b) ? mixed_kernel_of(b) : b);
  }
  for (size_t i = 0; i < items.size(); i++) {
    const uint32_t kind = progs[i].kernel, mixed = mixed_kernel_of(kind);
    const Key key{mixed ? mixed : kind, one_kernel ? 0u : workgroups_per_cu(kind, progs[i].lds_bytes)};
    size_t g = 0;
    while (g < keys.size() && !(keys[g] == key)) g++;
    if (g == keys.size()) {
      keys.push_back(key);
      members.emplace_back();
    }
    members[g].push_back((uint32_t)i);
  }
  auto slowness = [](uint32_t kind) {  // stream kinds by the time a 64 Ki-row block takes, longest first
    switch (kind >= KERNEL_STREAM_NONE + KERNEL_STREAM_NULLMAP ? kind - KERNEL_STREAM_NULLMAP : kind) {
      case KERNEL_STREAM_PSEUDO: return 0;
      case KERNEL_STREAM_DICT_GLB: return 1;
      case KERNEL_STREAM_DICT_LDS: return 2;
      case KERNEL_STREAM_ADD: return 3;
      default: return 4;
    }
  };
  std::vector<LaunchGroup> groups(keys.size());
  for (size_t g = 0; g < keys.size(); g++) {
    std::vector<uint32_t>& m = members[g];
    LaunchGroup& lg = groups[g];
    if (keys[g].kernel >= KERNEL_KINDS)
      std::stable_sort(m.begin(), m.end(), [&](uint32_t a, uint32_t b) { return slowness(progs[a].kernel) < slowness(progs[b].kernel); });
    lg.kernel = keys[g].kernel;
    lg.threads = threads_of(progs[m[0]].kernel);
    lg.count = m.size();
    uint32_t last_kind = ~0u;
    for (size_t j = 0; j < m.size(); j++) {
      const uint32_t i = m[j], kind = progs[i].kernel;
      lg.lds = std::max(lg.lds, progs[i].lds_bytes);
      lg.weight += ((uint64_t)items[i].tuple_count * (kind_is_double(kind) ? 8 : 4) + 1) * launch_cost(kind, keys[g].per_cu);
      if (kind != last_kind) {
        lg.seg.first_block[lg.seg.n] = (uint32_t)j;
        lg.seg.kind[lg.seg.n++] = kind >= KERNEL_STREAM_NONE + KERNEL_STREAM_NULLMAP && !kernel_is_generic(kind) ? kind - KERNEL_STREAM_NULLMAP : kind;
        last_kind = kind;
      }
    }
    lg.seg.first_block[lg.seg.n] = (uint32_t)m.size();
    if (lg.seg.n == 1) lg.kernel = progs[m[0]].kernel;  // one kind only: its own kernel
  }
  std::vector<size_t> by_weight(groups.size());
  for (size_t g = 0; g < groups.size(); g++) by_weight[g] = g;
  std::stable_sort(by_weight.begin(), by_weight.end(), [&](size_t a, size_t b) { return groups[a].weight > groups[b].weight; });
  lo.order.reserve(items.size());
  for (size_t g : by_weight) {
    groups[g].begin = lo.order.size();
    for (uint32_t i : members[g])
      lo.order.push_back(LaunchEntry{i, items[i].prog_off, items[i].prog_bytes, launch_entry_check(i, items[i].prog_off, items[i].prog_bytes)});
    if (groups[g].kernel >= KERNEL_KINDS) stats.kernel_mask |= 1u << groups[g].kernel;
    lo.launches.push_back(groups[g]);
  }
  stats.launches = (uint32_t)lo.launches.size();
}

// Uploads the built programs and work items of a plan.
btr_status finish_plan(btr_gpu_ctx* ctx, std::vector<WorkItem>& items, std::vector<BuiltProgram>& progs,
                       btr_plan_stats stats, btr_gpu_plan** out) {
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  const double t_start = now_s();
  auto plan = new btr_gpu_plan();
  plan->device = ctx->device;
  plan->n_items = items.size();
  Layout lo;
  lay_out(items, progs, stats, lo);
  const uint32_t debug_flags = debug_flags_from_env();
  for (WorkItem& it : items) it.debug_flags = debug_flags;
  plan->launches = lo.launches;
  const std::vector<uint8_t>& blob = lo.blob;
  const uint64_t scratch = lo.scratch;
  auto cleanup = [&](btr_status st) {
    btr_gpu_plan_destroy(plan);
    return st;
  };
  if (blob.size() > 0xFFFFFFF0ull) return cleanup(fail(BTR_ERR_UNSUPPORTED, "plan programs exceed 4 GiB"));
  // The launch with the most estimated work comes first. When the others are a few odd chunks (a
  // column whose blocks almost all take one cascade) it runs on the caller's stream and they are
  // small side launches on lanes of the context: raised wave priority, dispatched ahead of the main
  // grid. When several launches carry real shares of the work (a relation-level plan over many
  // columns) they are dealt, longest first, each to the stream with the least work so far: the
  // caller's stream and three lanes of the context. The caller's stream gets the first, longest
  // launch, and so the longest chain of launches: that chain, which decides when the decode ends,
  // then has no event hop between streams in front of it or behind it, and back-to-back decodes
  // follow each other on it without a gap (a rank's eighth of the mixed relation: 0.610 ms against
  // 0.634 ms with every launch on a lane). The context's lanes were created one after the other
  // and sit on different hardware queues.
  {
    const size_t n = plan->launches.size();
    plan->lane.assign(n, 0);
    plan->done.assign(n, nullptr);
    plan->go.assign(n, nullptr);
    plan->balanced = n > 1 && plan->launches[1].weight * 16 >= plan->launches[0].weight;
    if (plan->balanced) {
      // slot 0 is the caller's stream, slots 1..3 are lanes 1..3 of the context
      uint64_t load[kColumnLanes] = {};
      for (size_t j = 0; j < n; j++) {
        uint32_t best = 0;
        for (uint32_t l = 1; l < kColumnLanes; l++)
          if (load[l] < load[best]) best = l;
        load[best] += plan->launches[j].weight;
        plan->lane[j] = best ? best + 1 : 0;
      }
    } else if (n > 1) {
      const LaunchGroup& main = plan->launches[0];
      for (size_t i = 0; i < items.size(); i++) items[i].flags |= ITEM_SIDE_LAUNCH;
      for (size_t k = 0; k < main.count; k++) items[lo.order[main.begin + k].item].flags &= ~(uint32_t)ITEM_SIDE_LAUNCH;
    }
  }
  hipError_t e;
  if (scratch) {
    if ((e = hipMalloc((void**)&plan->d_scratch, scratch)) != hipSuccess)
      return cleanup(hip_fail(e, "hipMalloc(scratch)"));
  }
  for (size_t i = 0; i < items.size(); i++)
    items[i].scratch = progs[i].scratch_bytes ? (uint64_t)(uintptr_t)plan->d_scratch + items[i].scratch : 0;
  if ((debug_flags & 8) && !items.empty()) {
    {  // the step kinds of the first item's program, in stamp order (stamp 3 follows the first of them)
      const uint8_t* prog = blob.data() + items[0].prog_off;
      ProgramHeader hdr;
      memcpy(&hdr, prog, sizeof(hdr));
      fprintf(stderr, "[btr debug] steps after the %u index walks (kind[*: runs on into the next]):", hdr.n_bp_steps);
      for (uint32_t k = hdr.n_bp_steps; k < hdr.n_steps; k++) {
        Step st;
        memcpy(&st, prog + sizeof(hdr) + k * sizeof(Step), sizeof(st));
        fprintf(stderr, " %u%s", st.kind, st.no_barrier ? "*" : "");
      }
      fprintf(stderr, "\n");
    }
    if ((e = hipMalloc((void**)&plan->d_debug, items.size() * kDebugSlots * 4)) != hipSuccess)
      return cleanup(hip_fail(e, "hipMalloc(debug)"));
    (void)hipMemset(plan->d_debug, 0, items.size() * kDebugSlots * 4);
  }
  for (size_t i = 0; i < items.size(); i++)
    items[i].debug_buf = plan->d_debug ? (uint64_t)(uintptr_t)(plan->d_debug + i * kDebugSlots) : 0;
  if (!items.empty()) {
    if ((e = hipMalloc((void**)&plan->d_items, items.size() * sizeof(WorkItem))) != hipSuccess)
      return cleanup(hip_fail(e, "hipMalloc(items)"));
    if ((e = hipMalloc((void**)&plan->d_programs, blob.size() + 16)) != hipSuccess)
      return cleanup(hip_fail(e, "hipMalloc(programs)"));
    plan->program_bytes = blob.size();
    if ((e = hipMalloc((void**)&plan->d_status, items.size() * 4)) != hipSuccess)
      return cleanup(hip_fail(e, "hipMalloc(status)"));
    if ((e = hipMemcpy(plan->d_items, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice)) != hipSuccess)
      return cleanup(hip_fail(e, "hipMemcpy(items)"));
    if ((e = hipMemcpy(plan->d_programs, blob.data(), blob.size(), hipMemcpyHostToDevice)) != hipSuccess)
      return cleanup(hip_fail(e, "hipMemcpy(programs)"));
    if ((e = hipMemset(plan->d_status, 0, items.size() * 4)) != hipSuccess)
      return cleanup(hip_fail(e, "hipMemset(status)"));
    if (plan->launches.size() > 1) {
      if ((e = hipEventCreateWithFlags(&plan->fork, hipEventDisableTiming)) != hipSuccess)
        return cleanup(hip_fail(e, "hipEventCreate"));
      for (size_t j = plan->balanced ? 0 : 1; j < plan->launches.size(); j++) {
        if ((e = hipEventCreateWithFlags(&plan->done[j], hipEventDisableTiming)) != hipSuccess)
          return cleanup(hip_fail(e, "hipEventCreate"));
        if (!plan->balanced && (e = hipEventCreateWithFlags(&plan->go[j], hipEventDisableTiming)) != hipSuccess)
          return cleanup(hip_fail(e, "hipEventCreate"));
      }
    }
    if ((e = hipMalloc((void**)&plan->d_order, lo.order.size() * sizeof(LaunchEntry))) != hipSuccess)
      return cleanup(hip_fail(e, "hipMalloc(order)"));
    if ((e = hipMemcpy(plan->d_order, lo.order.data(), lo.order.size() * sizeof(LaunchEntry), hipMemcpyHostToDevice)) != hipSuccess)
      return cleanup(hip_fail(e, "hipMemcpy(order)"));
  }
  // The status words are cleared by a memset that the host does not wait for. A decode on a stream
  // that does not order itself behind the default stream must still find the plan complete.
  if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return cleanup(hip_fail(e, "hipStreamSynchronize"));
  stats.upload_seconds = now_s() - t_start;
  plan->stats = stats;
  *out = plan;
  return BTR_OK;
}

btr_status create_plan(btr_gpu_ctx* ctx, const btr_chunk_desc* chunks, size_t n, btr_gpu_plan** out) {
  std::vector<WorkItem> items;
  std::vector<BuiltProgram> progs;
  btr_plan_stats stats{};
  const double t0 = now_s();
  btr_status st;
  try {
    st = compile_chunks(chunks, n, items, progs, stats);
  } catch (const std::bad_alloc&) {
    return fail(BTR_ERR_OUT_OF_MEMORY, "out of host memory while compiling the plan");
  }
  if (st != BTR_OK) return st;
  stats.compile_seconds = now_s() - t0;
  return finish_plan(ctx, items, progs, stats, out);
}

// Chunk descriptors of chunks [first, first + count) of a column part image: outputs back to back
// from dev_dst (every chunk start 16-byte aligned), nullmaps back to back by rows.
btr_status part_descs(const void* host_part, const void* dev_part, size_t bytes, uint32_t first, uint32_t count,
                      void* dev_dst, size_t dst_bytes, uint8_t* dev_nullmap_dst, size_t nullmap_bytes,
                      std::vector<btr_chunk_desc>& descs) {
  uint32_t total;
  btr_status st = btr_part_chunk_count(host_part, bytes, &total);
  if (st != BTR_OK) return st;
  if (first > total || count > total - first) return fail(BTR_ERR_INVALID_ARGUMENT, "chunk range outside the part");
  descs.resize(count);
  size_t out_pos = 0, row_pos = 0;
  for (uint32_t k = 0; k < count; k++) {
    size_t off, len;
    if ((st = btr_part_chunk(host_part, bytes, first + k, &off, &len)) != BTR_OK) return st;
    btr_chunk_info info;
    if ((st = parse_chunk((const uint8_t*)host_part + off, len, &info)) != BTR_OK)
      return fail(st, "chunk " + std::to_string(first + k) + ": " + g_error);
    btr_chunk_desc& d = descs[k];
    d.host_src = (const uint8_t*)host_part + off;
    d.dev_src = (const uint8_t*)dev_part + off;
    d.src_bytes = len;
    d.dev_dst = (uint8_t*)dev_dst + out_pos;
    d.dst_bytes = info.decoded_bytes;
    d.dev_nullmap_dst = dev_nullmap_dst ? dev_nullmap_dst + row_pos : nullptr;
    out_pos += (info.decoded_bytes + 15) / 16 * 16;
    row_pos += info.tuple_count;
    if (out_pos - ((info.decoded_bytes + 15) / 16 * 16 - info.decoded_bytes) > dst_bytes)
      return fail(BTR_ERR_BUFFER_TOO_SMALL, "destination too small for the chunk range");
    if (dev_nullmap_dst && row_pos > nullmap_bytes) return fail(BTR_ERR_BUFFER_TOO_SMALL, "nullmap destination too small");
  }
  return BTR_OK;
}

}  // namespace

extern "C" {

const char* btr_last_error(void) { return g_error.c_str(); }
const char* btr_version(void) { return "btrblocks_amd 0.2 (gfx950)"; }
const char* btr_kernel_name(uint32_t k) {
  static const char* const names[KERNEL_KINDS] = {
      "btr_decode_int_kernel",      "btr_decode_dbl_kernel",      "btr_stream_none_kernel",  "btr_stream_add_kernel",
      "btr_stream_dict_lds_kernel", "btr_stream_dict_glb_kernel", "btr_stream_pseudo_kernel", "btr_stream_dictd_kernel",
      "btr_decode_int_pfor_kernel", "btr_decode_dbl_pfor_kernel",
      "btr_stream_none_nm_kernel", "btr_stream_add_nm_kernel", "btr_stream_dict_lds_nm_kernel",
      "btr_stream_dict_glb_nm_kernel", "btr_stream_pseudo_nm_kernel", "btr_stream_dictd_nm_kernel"};
  if (k == KERNEL_KINDS) return "btr_stream_mixed_kernel";
  if (k == KERNEL_KINDS + 1) return "btr_stream_mixed_nm_kernel";
  return k < KERNEL_KINDS ? names[k] : nullptr;
}

// ---- inspection -------------------------------------------------------------------------------
btr_status btr_chunk_inspect(const void* chunk, size_t bytes, btr_chunk_info* out) {
  if (!chunk || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  return parse_chunk((const uint8_t*)chunk, bytes, out);
}

btr_status btr_chunk_describe(const void* chunk, size_t bytes, char* out, size_t cap) {
  if (!chunk || !out || cap == 0) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  btr_chunk_info info;
  btr_status st = parse_chunk((const uint8_t*)chunk, bytes, &info);
  if (st != BTR_OK) return st;
  try {
    BuiltProgram bp;
    build_program(chunk_source((const uint8_t*)chunk, bytes, 0, info, false), bp);
    snprintf(out, cap, "%s", bp.description.c_str());
  } catch (const PlanError& e) {
    return fail((btr_status)e.status, e.what());
  } catch (const std::bad_alloc&) {
    return fail(BTR_ERR_OUT_OF_MEMORY, "out of host memory");
  } catch (const std::exception& e) {
    return fail(BTR_ERR_CORRUPT, e.what());
  }
  return BTR_OK;
}

btr_status btr_part_chunk_count(const void* part, size_t bytes, uint32_t* out) {
  if (!part || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (bytes < 4) return fail(BTR_ERR_CORRUPT, "part image shorter than its header");
  const uint32_t n = rd32((const uint8_t*)part + PART_NUM_CHUNKS);
  if ((size_t)n > (bytes - 4) / 4) return fail(BTR_ERR_CORRUPT, "chunk offset table past the end of the part");
  *out = n;
  return BTR_OK;
}

btr_status btr_part_chunk(const void* part, size_t bytes, uint32_t index, size_t* offset, size_t* length) {
  uint32_t n;
  btr_status st = btr_part_chunk_count(part, bytes, &n);
  if (st != BTR_OK) return st;
  if (index >= n) return fail(BTR_ERR_INVALID_ARGUMENT, "chunk index out of range");
  const uint8_t* p = (const uint8_t*)part;
  const size_t off = rd32(p + PART_OFFSETS + 4 * (size_t)index);
  const size_t end = index + 1 < n ? rd32(p + PART_OFFSETS + 4 * (size_t)(index + 1)) : bytes;
  if (off > end || end > bytes || end - off < CHUNK_DATA)
    return fail(BTR_ERR_CORRUPT, "chunk offsets are not increasing or run past the part");
  if (offset) *offset = off;
  if (length) *length = end - off;
  return BTR_OK;
}

btr_status btr_part_inspect(const void* part, size_t bytes, uint32_t first, uint32_t count, btr_chunk_info* infos,
                            uint64_t* offsets, uint64_t* lengths) {
  uint32_t total;
  btr_status st = btr_part_chunk_count(part, bytes, &total);
  if (st != BTR_OK) return st;
  if (first > total || count > total - first) return fail(BTR_ERR_INVALID_ARGUMENT, "chunk range outside the part");
  for (uint32_t k = 0; k < count; k++) {
    size_t off, len;
    if ((st = btr_part_chunk(part, bytes, first + k, &off, &len)) != BTR_OK) return st;
    if (offsets) offsets[k] = off;
    if (lengths) lengths[k] = len;
    if (infos && (st = parse_chunk((const uint8_t*)part + off, len, &infos[k])) != BTR_OK)
      return fail(st, "chunk " + std::to_string(first + k) + ": " + g_error);
  }
  return BTR_OK;
}

btr_status btr_datablock_inspect(const void* block, size_t bytes, uint32_t* tuple_count, uint32_t* column_count) {
  if (!block) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (bytes < DB_ATTRS) return fail(BTR_ERR_CORRUPT, "block shorter than its header");
  const uint8_t* p = (const uint8_t*)block;
  if (tuple_count) *tuple_count = rd32(p + DB_COUNT);
  if (column_count) *column_count = rd32(p + DB_COLUMNS);
  return BTR_OK;
}

btr_status btr_datablock_column(const void* block, size_t bytes, uint32_t column, btr_chunk_info* out) {
  if (!block || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  return parse_block_column((const uint8_t*)block, bytes, column, out, nullptr, nullptr);
}
// ---- compression (CPU) ----------------------------------------------------------------------------
size_t btr_compress_bound(uint8_t column_type, uint32_t n) {
  const size_t elem = column_type == COL_DOUBLE ? 8 : 4;
  // Worst case is the UNCOMPRESSED fallback plus envelope, a u32-array nullmap and alignment.
  return 64 + (size_t)n * elem + 8 + (size_t)n * 4;
}

btr_status btr_compress_chunk(uint8_t column_type, const void* values, const uint8_t* nullmap,
                              uint32_t n, const char* scheme_spec, const btr_compress_options* options,
                              void* out, size_t cap, size_t* out_bytes) {
  if ((!values && n) || !out || !out_bytes) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (column_type != COL_INTEGER && column_type != COL_DOUBLE)
    return fail(BTR_ERR_UNSUPPORTED, "only integer and double columns are supported");
  btrhost::Spec spec;
  std::string err;
  if (!btrhost::parse_spec(scheme_spec, column_type == COL_DOUBLE, spec, err))
    return fail(BTR_ERR_INVALID_ARGUMENT, err);
  try {
    std::vector<uint8_t> buf;
    btrhost::compress_chunk(column_type, values, nullmap, n, spec, to_options(options), buf);
    if (buf.size() > cap) return fail(BTR_ERR_BUFFER_TOO_SMALL, "output buffer too small");
    memcpy(out, buf.data(), buf.size());
    *out_bytes = buf.size();
  } catch (const std::exception& e) {
    return fail(BTR_ERR_UNSUPPORTED, e.what());
  }
  return BTR_OK;
}

btr_status btr_compress_datablock(const btr_column_input* columns, uint32_t column_count, uint32_t n,
                                  const btr_compress_options* options, void* out, size_t cap,
                                  size_t* out_bytes) {
  if (!columns || !out || !out_bytes) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<btrhost::ColumnInput> cols(column_count);
  for (uint32_t c = 0; c < column_count; c++) {
    cols[c].col_type = columns[c].column_type;
    cols[c].values = columns[c].values;
    cols[c].nullmap = columns[c].nullmap;
    std::string err;
    if (!btrhost::parse_spec(columns[c].scheme_spec, columns[c].column_type == COL_DOUBLE, cols[c].spec, err))
      return fail(BTR_ERR_INVALID_ARGUMENT, err);
  }
  try {
    std::vector<uint8_t> buf;
    btrhost::compress_datablock(cols, n, to_options(options), buf);
    if (buf.size() > cap) return fail(BTR_ERR_BUFFER_TOO_SMALL, "output buffer too small");
    memcpy(out, buf.data(), buf.size());
    *out_bytes = buf.size();
  } catch (const std::exception& e) {
    return fail(BTR_ERR_UNSUPPORTED, e.what());
  }
  return BTR_OK;
}

size_t btr_compress_string_dict_bound(uint64_t string_bytes, uint32_t n) {
  // pool (slots or symbol-table area + offsets + raw lengths) + raw codes + envelope, nullmap, padding
  return 256 + SDICT_FSST_HEADER + (size_t)string_bytes + 3 * ((size_t)n + 1) * 4 + (size_t)n * 4 + 8;
}

btr_status btr_compress_string_dict_chunk(const uint8_t* bytes, const uint32_t* offsets, uint32_t n,
                                          const uint8_t* nullmap, const char* codes_spec,
                                          const char* lengths_spec, int fsst_layout,
                                          const btr_compress_options* options, void* out, size_t cap,
                                          size_t* out_bytes) {
  if (!offsets || (!bytes && n && offsets[n] != offsets[0]) || !out || !out_bytes)
    return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  btrhost::StringDictOptions sopt;
  std::string err;
  if (!btrhost::parse_spec(codes_spec, false, sopt.codes_spec, err)) return fail(BTR_ERR_INVALID_ARGUMENT, err);
  if (!btrhost::parse_spec(lengths_spec, false, sopt.lengths_spec, err)) return fail(BTR_ERR_INVALID_ARGUMENT, err);
  sopt.fsst_layout = fsst_layout != 0;
  try {
    std::vector<uint8_t> buf;
    btrhost::compress_string_dict_chunk(bytes, offsets, n, nullmap, sopt, to_options(options), buf);
    if (buf.size() > cap) return fail(BTR_ERR_BUFFER_TOO_SMALL, "output buffer too small");
    memcpy(out, buf.data(), buf.size());
    *out_bytes = buf.size();
  } catch (const std::invalid_argument& e) {
    return fail(BTR_ERR_INVALID_ARGUMENT, e.what());
  } catch (const std::exception& e) {
    return fail(BTR_ERR_UNSUPPORTED, e.what());
  }
  return BTR_OK;
}

size_t btr_compress_column_bound(uint8_t column_type, uint64_t rows, uint32_t block_rows) {
  if (block_rows == 0) block_rows = 65536;
  const uint64_t chunks = (rows + block_rows - 1) / block_rows;
  return 16 + chunks * 4 + chunks * (btr_compress_bound(column_type, block_rows) + 16);
}

btr_status btr_compress_column(uint8_t column_type, const void* values, const uint8_t* nullmap,
                               uint64_t rows, uint32_t block_rows, const char* scheme_spec,
                               const btr_compress_options* options, int threads, void* out, size_t cap,
                               size_t* out_bytes) {
  if ((!values && rows) || !out || !out_bytes) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (column_type != COL_INTEGER && column_type != COL_DOUBLE)
    return fail(BTR_ERR_UNSUPPORTED, "only integer and double columns are supported");
  if (block_rows == 0) block_rows = 65536;
  btrhost::Spec spec;
  std::string err;
  if (!btrhost::parse_spec(scheme_spec, column_type == COL_DOUBLE, spec, err))
    return fail(BTR_ERR_INVALID_ARGUMENT, err);
  const btrhost::Options opt = to_options(options);
  const size_t elem = column_type == COL_DOUBLE ? 8 : 4;
  const uint64_t n_chunks = (rows + block_rows - 1) / block_rows;
  std::vector<std::vector<uint8_t>> chunks(n_chunks);
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  threads = std::max(1, std::min<int>(threads, (int)std::max<uint64_t>(1, n_chunks)));
  std::atomic<uint64_t> next{0};
  std::atomic<bool> failed{false};
  std::string worker_err;
  auto worker = [&]() {
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= n_chunks || failed) break;
      const uint64_t r0 = i * block_rows;
      const uint32_t n = (uint32_t)std::min<uint64_t>(block_rows, rows - r0);
      try {
        btrhost::compress_chunk(column_type, (const uint8_t*)values + r0 * elem,
                                nullmap ? nullmap + r0 : nullptr, n, spec, opt, chunks[i]);
      } catch (const std::exception& e) {
        if (!failed.exchange(true)) worker_err = e.what();
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  if (failed) return fail(BTR_ERR_UNSUPPORTED, worker_err);
  // column part image: {u32 num_chunks; u32 offsets[]} then 16-byte aligned chunks
  size_t total = 4 + 4 * n_chunks;
  for (auto& c : chunks) total = (total + 15) / 16 * 16 + c.size();
  if (total > 0xFFFFFFFFull) return fail(BTR_ERR_UNSUPPORTED, "column part image above 4 GiB; split the column");
  if (total > cap) return fail(BTR_ERR_BUFFER_TOO_SMALL, "output buffer too small");
  uint8_t* o = (uint8_t*)out;
  memset(o, 0, 4 + 4 * n_chunks);
  wr32(o, (uint32_t)n_chunks);
  size_t pos = 4 + 4 * n_chunks;
  for (uint64_t i = 0; i < n_chunks; i++) {
    const size_t aligned = (pos + 15) / 16 * 16;
    memset(o + pos, 0, aligned - pos);
    pos = aligned;
    wr32(o + PART_OFFSETS + 4 * i, (uint32_t)pos);
    memcpy(o + pos, chunks[i].data(), chunks[i].size());
    pos += chunks[i].size();
  }
  *out_bytes = pos;
  return BTR_OK;
}

// ---- GPU ----------------------------------------------------------------------------------------------
static const void* const kAllKernels[KERNEL_KINDS] = {
    (const void*)btr_decode_int_kernel,         (const void*)btr_decode_dbl_kernel,
    (const void*)btr_stream_none_kernel,        (const void*)btr_stream_add_kernel,
    (const void*)btr_stream_dict_lds_kernel,    (const void*)btr_stream_dict_glb_kernel,
    (const void*)btr_stream_pseudo_kernel,      (const void*)btr_stream_dictd_kernel,
    (const void*)btr_decode_int_pfor_kernel,    (const void*)btr_decode_dbl_pfor_kernel,
    (const void*)btr_stream_none_nm_kernel,     (const void*)btr_stream_add_nm_kernel,
    (const void*)btr_stream_dict_lds_nm_kernel, (const void*)btr_stream_dict_glb_nm_kernel,
    (const void*)btr_stream_pseudo_nm_kernel,   (const void*)btr_stream_dictd_nm_kernel};

btr_status btr_gpu_ctx_create(int device, btr_gpu_ctx** out) {
  if (!out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(BTR_ERR_DEVICE, "no HIP device available: decoding requires an MI355X (gfx950) GPU");
  if (device < 0 || device >= count || device >= kMaxDevices) return fail(BTR_ERR_INVALID_ARGUMENT, "device index out of range");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  for (const void* k : kAllKernels) {
    hipFuncAttributes attr;
    e = hipFuncGetAttributes(&attr, k);
    // blocks with many rows need more than the default 64 KiB of dynamic LDS
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCapBytes);
    if (e != hipSuccess) break;
  }
  if (e != hipSuccess)
    return fail(BTR_ERR_DEVICE, std::string("decode kernel has no code object for this device (built for gfx950): ") +
                                    hipGetErrorString(e));
  // Once per device, with that device current: the address of this GPU's copy of the static
  // area, its pinned staging mirror and the event that guards the mirror.
  StaticArea* area = &g_static_area[device];
  {
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!area->dev) {
      unsigned char* ptr = nullptr;
      if ((e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(g_static_scratch))) != hipSuccess || !ptr)
        return hip_fail(e, "hipGetSymbolAddress(static scratch)");
      unsigned char* staging = nullptr;
      if ((e = hipHostMalloc((void**)&staging, kStaticTablesOff, hipHostMallocDefault)) != hipSuccess)
        return hip_fail(e, "hipHostMalloc(staging)");
      hipEvent_t ev = nullptr;
      if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) {
        (void)hipHostFree(staging);
        return hip_fail(e, "hipEventCreate");
      }
      area->staging = staging;
      area->uploaded = ev;
      area->dev = ptr;
    }
  }
  btr_gpu_ctx* ctx = new btr_gpu_ctx;
  ctx->device = device;
  ctx->area = area;
  if ((e = hipEventCreateWithFlags(&ctx->lanes_fork, hipEventDisableTiming)) == hipSuccess) {
    for (uint32_t k = 0; k < kColumnLanes && e == hipSuccess; k++) {
      e = hipStreamCreateWithFlags(&ctx->lane[k], hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->lane_done[k], hipEventDisableTiming);
    }
    for (uint32_t k = 0; k < kPipeStreams && e == hipSuccess; k++)
      e = hipStreamCreateWithFlags(&ctx->pipe[k], hipStreamNonBlocking);
  }
  if (e != hipSuccess) {
    delete ctx;
    return hip_fail(e, "hipStreamCreate(context streams)");
  }
  *out = ctx;
  return BTR_OK;
}

void btr_gpu_ctx_destroy(btr_gpu_ctx* ctx) { delete ctx; }

btr_status btr_gpu_plan_create(btr_gpu_ctx* ctx, const btr_chunk_desc* chunks, size_t n, btr_gpu_plan** out) {
  if (!ctx || (!chunks && n) || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  return create_plan(ctx, chunks, n, out);
}

btr_status btr_gpu_plan_create_part(btr_gpu_ctx* ctx, const void* host_part, const void* dev_part, size_t bytes,
                                    uint32_t first_chunk, uint32_t chunk_count, void* dev_dst, size_t dst_bytes,
                                    uint8_t* dev_nullmap_dst, size_t nullmap_bytes, btr_gpu_plan** out) {
  if (!ctx || !host_part || !dev_part || (!dev_dst && chunk_count) || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<btr_chunk_desc> descs;
  btr_status st;
  try {
    st = part_descs(host_part, dev_part, bytes, first_chunk, chunk_count, dev_dst, dst_bytes, dev_nullmap_dst,
                    nullmap_bytes, descs);
  } catch (const std::bad_alloc&) {
    return fail(BTR_ERR_OUT_OF_MEMORY, "out of host memory");
  }
  if (st != BTR_OK) return st;
  return create_plan(ctx, descs.data(), descs.size(), out);
}

btr_status btr_gpu_plan_create_datablock(btr_gpu_ctx* ctx, const void* host_block, const void* dev_block,
                                         size_t bytes, void* const* dsts, uint8_t* const* nullmap_dsts,
                                         uint32_t column_count, btr_gpu_plan** out) {
  if (!ctx || !host_block || !dev_block || !dsts || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  const uint8_t* hb = (const uint8_t*)host_block;
  uint32_t n, cols;
  btr_status st = btr_datablock_inspect(host_block, bytes, &n, &cols);
  if (st != BTR_OK) return st;
  if (cols != column_count) return fail(BTR_ERR_INVALID_ARGUMENT, "column_count does not match the block");
  if (bytes > 0xFFFFFFFFull) return fail(BTR_ERR_UNSUPPORTED, "block above 4 GiB");
  const double t0 = now_s();
  std::vector<WorkItem> items(cols);
  std::vector<BuiltProgram> progs(cols);
  btr_plan_stats stats{};
  stats.stored_bytes = bytes;
  stats.needed_bytes = DB_ATTRS + (uint64_t)DB_ATTR_BYTES * cols;
  for (uint32_t c = 0; c < cols; c++) {
    btr_chunk_info info;
    uint32_t off, nm;
    st = parse_block_column(hb, bytes, c, &info, &off, &nm);
    if (st != BTR_OK) return st;
    if (info.column_type != COL_INTEGER && info.column_type != COL_DOUBLE)
      return fail(BTR_ERR_UNSUPPORTED, "column " + std::to_string(c) + ": only integer and double columns are decoded");
    if (!dsts[c]) return fail(BTR_ERR_INVALID_ARGUMENT, "column " + std::to_string(c) + ": null destination");
    if ((uint64_t)(uintptr_t)dsts[c] % 8 != 0)
      return fail(BTR_ERR_MISALIGNED, "column " + std::to_string(c) + ": destination must be 8-byte aligned");
    ColumnSource cs{};
    cs.host = hb;
    cs.bytes = bytes;
    cs.dev_addr = (uint64_t)(uintptr_t)dev_block;
    cs.payload_off = off;
    cs.nullmap_off = nm;
    cs.column_type = info.column_type;
    cs.scheme = info.scheme;
    cs.nullmap_type = info.nullmap_type;
    cs.tuple_count = n;
    cs.want_nullmap = nullmap_dsts && nullmap_dsts[c];
    try {
      build_program(cs, progs[c]);
    } catch (const PlanError& e) {
      return fail((btr_status)e.status, "column " + std::to_string(c) + ": " + e.what());
    } catch (const std::bad_alloc&) {
      return fail(BTR_ERR_OUT_OF_MEMORY, "out of host memory while compiling the decode program");
    } catch (const std::exception& e) {
      return fail(BTR_ERR_CORRUPT, "column " + std::to_string(c) + ": " + e.what());
    }
    WorkItem& it = items[c];
    it.src = (uint64_t)(uintptr_t)dev_block;
    it.dst = (uint64_t)(uintptr_t)dsts[c];
    it.nullmap_dst = cs.want_nullmap ? (uint64_t)(uintptr_t)nullmap_dsts[c] : 0;
    it.tuple_count = n;
    it.src_bytes = (uint32_t)bytes;
    stats.tuples += n;
    stats.decoded_bytes += info.decoded_bytes;
    if (cs.want_nullmap) stats.nullmap_bytes += n;
  }
  stats.compile_seconds = now_s() - t0;
  return finish_plan(ctx, items, progs, stats, out);
}

btr_status btr_string_dict_inspect(const void* chunk, size_t bytes, btr_string_dict_info* out) {
  if (!chunk || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  const uint8_t* p = (const uint8_t*)chunk;
  if (bytes < CHUNK_DATA) return fail(BTR_ERR_CORRUPT, "chunk shorter than its 12-byte envelope");
  if (p[CHUNK_COL_TYPE] != COL_STRING || p[CHUNK_SCHEME] != S_DICT)
    return fail(BTR_ERR_UNSUPPORTED, "not a string column chunk in the DICT scheme");
  if (bytes > 0xFFFFFFFFull) return fail(BTR_ERR_UNSUPPORTED, "chunk above 4 GiB");
  const size_t pool = (size_t)CHUNK_DATA + SDICT_DATA;
  if (bytes < pool) return fail(BTR_ERR_CORRUPT, "chunk shorter than the string dictionary header");
  const uint8_t* s = p + CHUNK_DATA;
  memset(out, 0, sizeof(*out));
  out->tuple_count = rd32(p + CHUNK_TUPLES);
  out->total_decompressed_size = rd32(s + SDICT_TOTAL_SIZE);
  out->num_codes = rd32(s + SDICT_NUM_CODES);
  out->use_fsst = s[SDICT_USE_FSST] != 0;
  out->use_rle_optimized_path = s[SDICT_USE_RLE] != 0;
  out->codes_scheme = s[SDICT_CODES_SCHEME];
  out->lengths_scheme = s[SDICT_LENGTHS_SCHEME];
  out->pool_offset = (uint32_t)pool;
  const size_t room = bytes - pool;
  auto rebase = [&](uint32_t field, uint32_t* to, const char* what) -> btr_status {
    const uint32_t off = rd32(s + field);
    if (off > room) return fail(BTR_ERR_CORRUPT, std::string(what) + " offset past the end of the chunk");
    *to = (uint32_t)(pool + off);
    return BTR_OK;
  };
  btr_status st;
  if ((st = rebase(SDICT_CODES_OFFSET, &out->codes_offset, "codes")) != BTR_OK) return st;
  if (out->use_fsst) {
    if ((st = rebase(SDICT_LENGTHS_OFFSET, &out->lengths_offset, "lengths")) != BTR_OK) return st;
    if ((st = rebase(SDICT_FSST_OFFSETS_OFFSET, &out->fsst_offsets_offset, "string offsets")) != BTR_OK) return st;
  }
  if (out->num_codes > out->tuple_count) return fail(BTR_ERR_CORRUPT, "more dictionary entries than rows");
  if (out->use_rle_optimized_path) {
    if (out->codes_scheme != I_RLE) return fail(BTR_ERR_CORRUPT, "RLE-optimised path without an RLE codes stream");
    const size_t node = out->codes_offset;
    if (bytes - node < RLE_DATA) return fail(BTR_ERR_CORRUPT, "RLE header of the codes past the end of the chunk");
    out->runs_count = rd32(p + node + RLE_RUNS);
    const uint32_t counts_off = rd32(p + node + RLE_COUNTS_OFFSET);
    if (out->runs_count > out->tuple_count) return fail(BTR_ERR_CORRUPT, "more runs than rows");
    if (counts_off > bytes - node - RLE_DATA) return fail(BTR_ERR_CORRUPT, "run lengths past the end of the chunk");
    out->run_values_scheme = p[node + RLE_VALUES_SCHEME];
    out->run_lengths_scheme = p[node + RLE_COUNTS_SCHEME];
    out->run_values_offset = (uint32_t)(node + RLE_DATA);
    out->run_lengths_offset = (uint32_t)(node + RLE_DATA + counts_off);
  }
  return BTR_OK;
}

btr_status btr_gpu_plan_create_string_dict(btr_gpu_ctx* ctx, const btr_substream_desc* subs, size_t n,
                                           btr_gpu_plan** out) {
  if (!ctx || (!subs && n) || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  const double t0 = now_s();
  std::vector<WorkItem> items(n);
  std::vector<BuiltProgram> progs(n);
  btr_plan_stats stats{};
  for (size_t i = 0; i < n; i++) {
    const btr_substream_desc& d = subs[i];
    const std::string where = "item " + std::to_string(i) + ": ";
    if (!d.host_src || !d.dev_src || !d.dev_dst) return fail(BTR_ERR_INVALID_ARGUMENT, where + "null pointer");
    btr_string_dict_info info;
    btr_status st = btr_string_dict_inspect(d.host_src, d.src_bytes, &info);
    if (st != BTR_OK) return fail(st, where + g_error);
    ColumnSource cs{};
    cs.host = (const uint8_t*)d.host_src;
    cs.bytes = d.src_bytes;
    cs.dev_addr = (uint64_t)(uintptr_t)d.dev_src;
    cs.column_type = COL_INTEGER;
    cs.nullmap_type = NM_ALLONES;  // the sub-streams are compressed without a nullmap
    cs.want_nullmap = false;
    switch (d.which) {
      case BTR_SUBSTREAM_CODES:
        cs.payload_off = info.codes_offset;
        cs.scheme = info.codes_scheme;
        cs.tuple_count = info.tuple_count;
        break;
      case BTR_SUBSTREAM_LENGTHS:
        if (!info.use_fsst) return fail(BTR_ERR_INVALID_ARGUMENT, where + "a chunk without FSST has no lengths stream");
        cs.payload_off = info.lengths_offset;
        cs.scheme = info.lengths_scheme;
        cs.tuple_count = info.num_codes;
        break;
      case BTR_SUBSTREAM_RUN_VALUES:
      case BTR_SUBSTREAM_RUN_LENGTHS:
        if (!info.use_rle_optimized_path)
          return fail(BTR_ERR_INVALID_ARGUMENT, where + "the chunk does not use the RLE-optimised path");
        cs.payload_off = d.which == BTR_SUBSTREAM_RUN_VALUES ? info.run_values_offset : info.run_lengths_offset;
        cs.scheme = d.which == BTR_SUBSTREAM_RUN_VALUES ? info.run_values_scheme : info.run_lengths_scheme;
        cs.tuple_count = info.runs_count;
        break;
      default: return fail(BTR_ERR_INVALID_ARGUMENT, where + "unknown sub-stream");
    }
    if (d.dst_bytes < (uint64_t)cs.tuple_count * 4) return fail(BTR_ERR_BUFFER_TOO_SMALL, where + "destination too small");
    if ((uint64_t)(uintptr_t)d.dev_dst % 8 != 0) return fail(BTR_ERR_MISALIGNED, where + "destination must be 8-byte aligned");
    try {
      build_program(cs, progs[i]);
    } catch (const PlanError& e) {
      return fail((btr_status)e.status, where + e.what());
    } catch (const std::bad_alloc&) {
      return fail(BTR_ERR_OUT_OF_MEMORY, "out of host memory while compiling the decode program");
    } catch (const std::exception& e) {
      return fail(BTR_ERR_CORRUPT, where + e.what());
    }
    WorkItem& it = items[i];
    it.src = (uint64_t)(uintptr_t)d.dev_src;
    it.dst = (uint64_t)(uintptr_t)d.dev_dst;
    it.nullmap_dst = 0;
    it.tuple_count = cs.tuple_count;
    it.src_bytes = (uint32_t)d.src_bytes;
    stats.tuples += cs.tuple_count;
    stats.decoded_bytes += (uint64_t)cs.tuple_count * 4;
    stats.stored_bytes += d.src_bytes;
    stats.needed_bytes += CHUNK_DATA + SDICT_DATA;
  }
  stats.compile_seconds = now_s() - t0;
  return finish_plan(ctx, items, progs, stats, out);
}

void btr_gpu_plan_destroy(btr_gpu_plan* plan) {
  if (!plan) return;
  DeviceGuard guard(plan->device);
  if (plan->d_items) (void)hipFree(plan->d_items);
  if (plan->d_programs) (void)hipFree(plan->d_programs);
  if (plan->d_status) (void)hipFree(plan->d_status);
  if (plan->d_debug) (void)hipFree(plan->d_debug);
  if (plan->d_order) (void)hipFree(plan->d_order);
  if (plan->d_scratch) (void)hipFree(plan->d_scratch);
  for (hipEvent_t ev : plan->done)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : plan->go)
    if (ev) (void)hipEventDestroy(ev);
  if (plan->fork) (void)hipEventDestroy(plan->fork);
  delete plan;
}

btr_status btr_gpu_plan_stats(const btr_gpu_plan* plan, btr_plan_stats* out) {
  if (!plan || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  *out = plan->stats;
  return BTR_OK;
}

using DecodeKernel = void (*)(const WorkItem*, const LaunchEntry*, const uint8_t*, uint32_t*);
static DecodeKernel kernel_of(uint32_t kind) {
  static const DecodeKernel kernels[KERNEL_KINDS] = {btr_decode_int_kernel,      btr_decode_dbl_kernel,
                                                     btr_stream_none_kernel,     btr_stream_add_kernel,
                                                     btr_stream_dict_lds_kernel, btr_stream_dict_glb_kernel,
                                                     btr_stream_pseudo_kernel,   btr_stream_dictd_kernel,
                                                     btr_decode_int_pfor_kernel, btr_decode_dbl_pfor_kernel,
                                                     btr_stream_none_nm_kernel,  btr_stream_add_nm_kernel,
                                                     btr_stream_dict_lds_nm_kernel, btr_stream_dict_glb_nm_kernel,
                                                     btr_stream_pseudo_nm_kernel, btr_stream_dictd_nm_kernel};
  return kernels[kind];
}
// Enqueues one launch of a layout. `order` is the whole launch order on the device.
static hipError_t enqueue_launch(const LaunchGroup& lg, hipStream_t on, const WorkItem* d_items, const LaunchEntry* d_order,
                                 const uint8_t* d_programs, uint32_t* d_status) {
  if (lg.kernel == kMixedStream)
    hipLaunchKernelGGL(btr_stream_mixed_kernel, dim3((unsigned)lg.count), dim3(lg.threads), lg.lds, on, d_items,
                       d_order + lg.begin, d_programs, d_status, lg.seg);
  else if (lg.kernel == kMixedStreamNm)
    hipLaunchKernelGGL(btr_stream_mixed_nm_kernel, dim3((unsigned)lg.count), dim3(lg.threads), lg.lds, on, d_items,
                       d_order + lg.begin, d_programs, d_status, lg.seg);
  else
    hipLaunchKernelGGL(kernel_of(lg.kernel), dim3((unsigned)lg.count), dim3(lg.threads), lg.lds, on, d_items,
                       d_order + lg.begin, d_programs, d_status);
  return hipGetLastError();
}

btr_status btr_gpu_decode(btr_gpu_ctx* ctx, btr_gpu_plan* plan, void* hip_stream) {
  if (!ctx || !plan) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (plan->n_items == 0) return BTR_OK;
  DeviceGuard guard(plan->device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  hipStream_t stream = (hipStream_t)hip_stream;
  auto launch = [&](size_t j, hipStream_t on) {
    return enqueue_launch(plan->launches[j], on, plan->d_items, plan->d_order, plan->d_programs, plan->d_status);
  };
  const size_t n = plan->launches.size();
  if (n <= 1) {
    if (n) HIP_TRY(launch(0, stream));
    return BTR_OK;
  }
  std::lock_guard<std::recursive_mutex> lock(ctx->streams_mutex);
  HIP_TRY(hipEventRecord(plan->fork, stream));
  if (plan->balanced) {
    // several large launches on the caller's stream and lanes of the context (btr_gpu_plan::lane); one join per
    // lane, behind its last launch
    bool forked[kColumnLanes + 1] = {};
    size_t last_on[kColumnLanes + 1];
    for (size_t j = 0; j < n; j++) last_on[plan->lane[j]] = j;
    auto stream_of = [&](size_t j) { return plan->lane[j] ? ctx->lane[plan->lane[j] - 1] : stream; };
    for (size_t j = 0; j < n; j++) {
      const uint32_t l = plan->lane[j];
      hipStream_t on = stream_of(j);
      if (on != stream && !forked[l]) {  // (the caller may be decoding on one of the lanes: btr_gpu_decode_columns)
        HIP_TRY(hipStreamWaitEvent(on, plan->fork, 0));
        forked[l] = true;
      }
      HIP_TRY(launch(j, on));
      if (on != stream && last_on[l] == j) HIP_TRY(hipEventRecord(plan->done[j], on));
    }
    for (size_t j = 0; j < n; j++)
      if (stream_of(j) != stream && last_on[plan->lane[j]] == j) HIP_TRY(hipStreamWaitEvent(stream, plan->done[j], 0));
    return BTR_OK;
  }
  // The small launches fork from the caller's stream first, the main one follows on the stream
  // itself, then the stream waits for the side streams. The main launch waits for an event that each
  // side stream records right before its own launch: the few workgroups of a side launch (which may
  // need more LDS than the main launch leaves free on any CU) are then dispatched onto an empty GPU
  // instead of queueing for a slot until the main grid has been handed out.
  for (size_t j = 1; j < n; j++) {
    // lanes 1..3, as for the large launches above: lane 0 was seen to share its hardware queue with the
    // default stream, and a side launch in the caller's own queue runs before the main launch, not beside it
    hipStream_t on = ctx->lane[1 + ctx->next_side++ % (kColumnLanes - 1)];
    HIP_TRY(hipStreamWaitEvent(on, plan->fork, 0));
    HIP_TRY(hipEventRecord(plan->go[j], on));
    HIP_TRY(launch(j, on));
    HIP_TRY(hipEventRecord(plan->done[j], on));
  }
  for (size_t j = 1; j < n; j++) HIP_TRY(hipStreamWaitEvent(stream, plan->go[j], 0));
  HIP_TRY(launch(0, stream));
  for (size_t j = 1; j < n; j++) HIP_TRY(hipStreamWaitEvent(stream, plan->done[j], 0));
  return BTR_OK;
}

btr_status btr_gpu_decode_columns(btr_gpu_ctx* ctx, btr_gpu_plan* const* plans, size_t n, void* hip_stream) {
  if (!ctx || (!plans && n)) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  for (size_t i = 0; i < n; i++) {
    if (!plans[i]) return fail(BTR_ERR_INVALID_ARGUMENT, "null plan");
    if (plans[i]->device != ctx->device) return fail(BTR_ERR_INVALID_ARGUMENT, "plan " + std::to_string(i) + " belongs to another device");
  }
  if (n == 0) return BTR_OK;
  if (n == 1) return btr_gpu_decode(ctx, plans[0], hip_stream);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  hipStream_t stream = (hipStream_t)hip_stream;
  std::lock_guard<std::recursive_mutex> lock(ctx->streams_mutex);
  // plan i goes to slot i mod 4: slot 0 is the caller's stream itself, slots 1..3 are lanes 1..3 of the
  // context (lane 0 shares its hardware queue with the default stream, see btr_gpu_decode)
  const uint32_t used = (uint32_t)std::min<size_t>(n, kColumnLanes);
  HIP_TRY(hipEventRecord(ctx->lanes_fork, stream));
  for (uint32_t k = 1; k < used; k++) HIP_TRY(hipStreamWaitEvent(ctx->lane[k], ctx->lanes_fork, 0));
  for (size_t i = 0; i < n; i++) {
    const uint32_t slot = (uint32_t)(i % kColumnLanes);
    const btr_status st = btr_gpu_decode(ctx, plans[i], slot ? ctx->lane[slot] : stream);
    if (st != BTR_OK) return st;
  }
  for (uint32_t k = 1; k < used; k++) {
    HIP_TRY(hipEventRecord(ctx->lane_done[k], ctx->lane[k]));
    HIP_TRY(hipStreamWaitEvent(stream, ctx->lane_done[k], 0));
  }
  return BTR_OK;
}

btr_status btr_gpu_decode_chunks_noalloc(btr_gpu_ctx* ctx, const btr_chunk_desc* chunks, size_t n, void* hip_stream,
                                         uint32_t* status_or) {
  if (!ctx || (!chunks && n)) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (status_or) *status_or = 0;
  if (n == 0) return BTR_OK;
  if (n > kStaticMaxChunks)
    return fail(BTR_ERR_INVALID_ARGUMENT, "at most " + std::to_string(kStaticMaxChunks) + " chunks per allocation-free call");
  StaticArea* area = ctx->area;
  if (!area || !area->dev) return fail(BTR_ERR_DEVICE, "static decode area not initialised for this device");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  std::vector<WorkItem> items;
  std::vector<BuiltProgram> progs;
  btr_plan_stats stats{};
  Layout lo;
  try {
    const btr_status st = compile_chunks(chunks, n, items, progs, stats, 1);  // on the calling thread: no threads are started
    if (st != BTR_OK) return st;
    lay_out(items, progs, stats, lo);
  } catch (const std::bad_alloc&) {
    return fail(BTR_ERR_OUT_OF_MEMORY, "out of host memory while compiling the plan");
  }
  if (kStaticProgOff + lo.blob.size() > kStaticTablesOff)
    return fail(BTR_ERR_BUFFER_TOO_SMALL, "decode programs of these chunks exceed the static area (" +
                                              std::to_string(kStaticTablesOff - kStaticProgOff) + " bytes)");
  if (lo.scratch > kStaticBytes - kStaticTablesOff)
    return fail(BTR_ERR_BUFFER_TOO_SMALL, "decode tables of these chunks exceed the static area (" +
                                              std::to_string(kStaticBytes - kStaticTablesOff) + " bytes)");
  unsigned char* const dev = area->dev;  // this GPU's copy of the static array
  for (size_t i = 0; i < n; i++) {
    items[i].scratch = progs[i].scratch_bytes ? (uint64_t)(uintptr_t)(dev + kStaticTablesOff) + items[i].scratch : 0;
    items[i].debug_flags = 0;
    items[i].debug_buf = 0;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  // the previous call's upload must have left the pinned mirror before it is refilled
  HIP_TRY(hipEventSynchronize(area->uploaded));
  memcpy(area->staging + kStaticItemsOff, items.data(), n * sizeof(WorkItem));
  memcpy(area->staging + kStaticOrderOff, lo.order.data(), n * sizeof(LaunchEntry));
  memset(area->staging + kStaticStatusOff, 0, kStaticMaxChunks * 4);  // status words start at zero
  memcpy(area->staging + kStaticProgOff, lo.blob.data(), lo.blob.size());
  HIP_TRY(hipMemcpyAsync(dev, area->staging, kStaticProgOff + lo.blob.size(), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipEventRecord(area->uploaded, stream));
  const WorkItem* d_items = reinterpret_cast<const WorkItem*>(dev + kStaticItemsOff);
  const LaunchEntry* d_order = reinterpret_cast<const LaunchEntry*>(dev + kStaticOrderOff);
  uint32_t* d_status = reinterpret_cast<uint32_t*>(dev + kStaticStatusOff);
  const uint8_t* d_programs = dev + kStaticProgOff;
  for (const LaunchGroup& lg : lo.launches)  // back to back on the stream
    HIP_TRY(enqueue_launch(lg, stream, d_items, d_order, d_programs, d_status));
  if (!status_or) return BTR_OK;
  HIP_TRY(hipStreamSynchronize(stream));
  uint32_t words[kStaticMaxChunks];
  HIP_TRY(hipMemcpy(words, d_status, n * 4, hipMemcpyDeviceToHost));
  uint32_t any = 0;
  size_t first = n;
  for (size_t i = 0; i < n; i++) {
    if (words[i] && first == n) first = i;
    any |= words[i];
  }
  *status_or = any;
  if (any) return fail(BTR_ERR_CORRUPT, "malformed compressed stream in chunk " + std::to_string(first));
  return BTR_OK;
}

btr_status btr_gpu_plan_check(btr_gpu_ctx* ctx, btr_gpu_plan* plan, void* hip_stream, uint32_t* status_or,
                              uint64_t* first_bad) {
  if (!ctx || !plan) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGuard guard(plan->device);
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  plan->h_status.resize(plan->n_items);
  if (plan->n_items)
    HIP_TRY(hipMemcpy(plan->h_status.data(), plan->d_status, plan->n_items * 4, hipMemcpyDeviceToHost));
  if (plan->d_debug && plan->n_items) {
    // timing diagnostics: per slot, mean offset from the item's first stamp; and when items started
    std::vector<uint32_t> ts(plan->n_items * kDebugSlots);
    HIP_TRY(hipMemcpy(ts.data(), plan->d_debug, ts.size() * 4, hipMemcpyDeviceToHost));
    uint32_t t0 = ~0u;
    for (size_t i = 0; i < plan->n_items; i++)
      if (ts[i * kDebugSlots]) t0 = std::min(t0, ts[i * kDebugSlots]);
    fprintf(stderr, "[btr debug] %zu items; mean us since item start per stamp:", plan->n_items);
    for (uint32_t k = 1; k < kDebugSlots; k++) {
      double sum = 0;
      size_t cnt = 0;
      for (size_t i = 0; i < plan->n_items; i++)
        if (ts[i * kDebugSlots] && ts[i * kDebugSlots + k]) {
          sum += (double)(uint32_t)(ts[i * kDebugSlots + k] - ts[i * kDebugSlots]) * 0.01;
          cnt++;
        }
      if (cnt) fprintf(stderr, " [%u] %.2f", k, sum / cnt);
    }
    std::vector<uint32_t> starts;
    for (size_t i = 0; i < plan->n_items; i++)
      if (ts[i * kDebugSlots]) starts.push_back(ts[i * kDebugSlots] - t0);
    std::sort(starts.begin(), starts.end());
    fprintf(stderr, "\n[btr debug] item start times (us), deciles:");
    for (int d = 0; d <= 10 && !starts.empty(); d++) fprintf(stderr, " %.1f", starts[(starts.size() - 1) * d / 10] * 0.01);
    fprintf(stderr, "\n");
    for (const LaunchGroup& lg : plan->launches) {
      int per_cu = 0;
      const void* fn = lg.kernel == kMixedStream     ? reinterpret_cast<const void*>(btr_stream_mixed_kernel)
                       : lg.kernel == kMixedStreamNm ? reinterpret_cast<const void*>(btr_stream_mixed_nm_kernel)
                                                     : reinterpret_cast<const void*>(kernel_of(lg.kernel));
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)lg.threads, lg.lds);
      fprintf(stderr, "[btr debug] %s: %zu workgroups, %u bytes of LDS, %d workgroups per CU\n", btr_kernel_name(lg.kernel),
              lg.count, lg.lds, per_cu);
    }
  }
  uint32_t any = 0;
  uint64_t first = ~0ull;
  for (size_t i = 0; i < plan->n_items; i++) {
    if (plan->h_status[i] && first == ~0ull) first = i;
    any |= plan->h_status[i];
  }
  // ST_PROGRAM is the one bit the kernels do not reset at the start of a decode: it is cleared here,
  // once it has been read
  if (any & ST_PROGRAM) HIP_TRY(hipMemset(plan->d_status, 0, plan->n_items * 4));
  if (status_or) *status_or = any;
  if (first_bad) *first_bad = first;
  if (any)
    return fail(BTR_ERR_CORRUPT, "malformed compressed stream in chunk " + std::to_string(first) +
                                     " (status bits 0x" + [&] { char b[16]; snprintf(b, sizeof b, "%x", any); return std::string(b); }() + ")");
  return BTR_OK;
}

// ---- test hooks ---------------------------------------------------------------------------------------
uint32_t btr_debug_launch_entry_check(uint32_t item, uint32_t prog_off, uint32_t prog_bytes) {
  return launch_entry_check(item, prog_off, prog_bytes);
}

btr_status btr_debug_launch_layout(const btr_chunk_desc* chunks, size_t n, char* text, size_t text_bytes) {
  if ((!chunks && n) || !text || !text_bytes) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<WorkItem> items;
  std::vector<BuiltProgram> progs;
  btr_plan_stats stats{};
  Layout lo;
  try {
    const btr_status st = compile_chunks(chunks, n, items, progs, stats, 1);
    if (st != BTR_OK) return st;
    lay_out(items, progs, stats, lo);
  } catch (const std::bad_alloc&) {
    return fail(BTR_ERR_OUT_OF_MEMORY, "out of host memory while compiling the plan");
  }
  std::string out;
  for (const LaunchGroup& lg : lo.launches) {
    out += std::string(btr_kernel_name(lg.kernel)) + " threads=" + std::to_string(lg.threads) + " lds=" + std::to_string(lg.lds) +
           " chunks=" + std::to_string(lg.count) + " [";
    for (uint32_t k = 0; k < lg.seg.n; k++)
      out += (k ? " " : "") + std::to_string(lg.seg.first_block[k]) + ":" + btr_kernel_name(lg.seg.kind[k]);
    out += "] items=";
    for (size_t k = 0; k < lg.count; k++) out += (k ? "," : "") + std::to_string(lo.order[lg.begin + k].item);
    out += "\n";
  }
  if (out.size() + 1 > text_bytes) return fail(BTR_ERR_BUFFER_TOO_SMALL, "launch description needs " + std::to_string(out.size() + 1) + " bytes");
  memcpy(text, out.c_str(), out.size() + 1);
  return BTR_OK;
}

btr_status btr_gpu_plan_debug_access(btr_gpu_plan* plan, uint32_t region, size_t offset, void* host, size_t bytes,
                                     int write) {
  if (!plan || (!host && bytes)) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGuard guard(plan->device);
  uint8_t* base = nullptr;
  size_t size = 0;
  switch (region) {
    case 0: base = (uint8_t*)plan->d_items, size = plan->n_items * sizeof(WorkItem); break;
    case 1: base = (uint8_t*)plan->d_order, size = plan->n_items * sizeof(LaunchEntry); break;
    case 2: base = plan->d_programs, size = plan->program_bytes; break;
    default: return fail(BTR_ERR_INVALID_ARGUMENT, "unknown region");
  }
  if (offset > size || bytes > size - offset) return fail(BTR_ERR_INVALID_ARGUMENT, "range outside the region");
  HIP_TRY(hipDeviceSynchronize());
  if (bytes) HIP_TRY(write ? hipMemcpy(base + offset, host, bytes, hipMemcpyHostToDevice)
                           : hipMemcpy(host, base + offset, bytes, hipMemcpyDeviceToHost));
  return BTR_OK;
}

btr_status btr_gpu_decode_chunk_host(btr_gpu_ctx* ctx, const void* chunk, size_t bytes, void* dst,
                                     size_t dst_bytes, uint8_t* nullmap_dst) {
  if (!ctx || !chunk || !dst) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  btr_chunk_info info;
  btr_status st = parse_chunk((const uint8_t*)chunk, bytes, &info);
  if (st != BTR_OK) return st;
  if (dst_bytes < info.decoded_bytes) return fail(BTR_ERR_BUFFER_TOO_SMALL, "destination too small");
  DeviceGuard guard(ctx->device);
  uint8_t *d_src = nullptr, *d_dst = nullptr, *d_nm = nullptr;
  btr_gpu_plan* plan = nullptr;
  auto done = [&](btr_status s) {
    if (plan) btr_gpu_plan_destroy(plan);
    if (d_src) (void)hipFree(d_src);
    if (d_dst) (void)hipFree(d_dst);
    if (d_nm) (void)hipFree(d_nm);
    return s;
  };
  hipError_t e;
  if ((e = hipMalloc((void**)&d_src, bytes + 16)) != hipSuccess) return done(hip_fail(e, "hipMalloc"));
  if ((e = hipMalloc((void**)&d_dst, info.decoded_bytes + 16)) != hipSuccess) return done(hip_fail(e, "hipMalloc"));
  if (nullmap_dst && (e = hipMalloc((void**)&d_nm, info.tuple_count + 16)) != hipSuccess)
    return done(hip_fail(e, "hipMalloc"));
  if ((e = hipMemcpy(d_src, chunk, bytes, hipMemcpyHostToDevice)) != hipSuccess) return done(hip_fail(e, "hipMemcpy"));
  btr_chunk_desc d{chunk, d_src, bytes, d_dst, (size_t)info.decoded_bytes, d_nm};
  if ((st = btr_gpu_plan_create(ctx, &d, 1, &plan)) != BTR_OK) return done(st);
  if ((st = btr_gpu_decode(ctx, plan, nullptr)) != BTR_OK) return done(st);
  if ((st = btr_gpu_plan_check(ctx, plan, nullptr, nullptr, nullptr)) != BTR_OK) return done(st);
  if (info.decoded_bytes && (e = hipMemcpy(dst, d_dst, info.decoded_bytes, hipMemcpyDeviceToHost)) != hipSuccess)
    return done(hip_fail(e, "hipMemcpy"));
  if (nullmap_dst && info.tuple_count &&
      (e = hipMemcpy(nullmap_dst, d_nm, info.tuple_count, hipMemcpyDeviceToHost)) != hipSuccess)
    return done(hip_fail(e, "hipMemcpy"));
  return done(BTR_OK);
}

btr_status btr_gpu_malloc(btr_gpu_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGuard guard(ctx->device);
  hipError_t e = hipMalloc(out, bytes ? bytes : 1);
  if (e == hipErrorOutOfMemory) return fail(BTR_ERR_OUT_OF_MEMORY, "hipMalloc: out of device memory");
  HIP_TRY(e);
  return BTR_OK;
}
btr_status btr_gpu_free(btr_gpu_ctx* ctx, void* ptr) {
  if (!ctx) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGuard guard(ctx->device);
  HIP_TRY(hipFree(ptr));
  return BTR_OK;
}
btr_status btr_gpu_upload(btr_gpu_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
  if (!ctx) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGuard guard(ctx->device);
  if (bytes) HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
  return BTR_OK;
}
btr_status btr_gpu_download(btr_gpu_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
  if (!ctx) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGuard guard(ctx->device);
  if (bytes) HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
  return BTR_OK;
}
btr_status btr_gpu_synchronize(btr_gpu_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGuard guard(ctx->device);
  HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
  return BTR_OK;
}

}  // extern "C"

#include "pipeline.hip"
