// Streaming decode of column parts that start in host memory: the part files of a relation
// directory (btrblocks/storage/Chunk.cpp:250-295) as tools/conversion/decompression-speed.cpp:160-170
// feeds them to BtrReader objects. Included by c_api.hip (it uses that file's plan compiler and
// kernel table).
//
// A pipeline owns `slots` sets of device buffers and borrows the context's three pipeline streams
// (pipelines of one context share them). For part p, in slot p % slots:
//   host      header walk + program compilation of the part (the work BtrReader::readColumn does
//             per chunk, BtrReader.cpp:31-74), into the slot's pinned staging area
//   copy-in   part image and programs host -> device
//   decode    one launch per kernel kind over all chunks of the part
//   copy-out  decoded column, nullmap and status words device -> host
// The stages of one part are ordered by events; different parts overlap, so while part p decodes,
// part p + 1 is being copied in and compiled and part p - 1 is being copied out.

namespace {

struct PipeSlot {
  // device
  uint8_t* d_in = nullptr;
  size_t d_in_cap = 0;
  uint8_t* d_out = nullptr;
  size_t d_out_cap = 0;
  uint8_t* d_nm = nullptr;
  size_t d_nm_cap = 0;
  uint8_t* d_arena = nullptr;  // items | order | status | programs | table slices
  size_t d_arena_cap = 0;
  // pinned host
  uint8_t* h_in = nullptr;  // part image handed out by btr_gpu_pipeline_acquire
  size_t h_in_cap = 0;
  uint8_t* h_stage = nullptr;  // mirror of the uploaded head of the arena
  size_t h_stage_cap = 0;
  uint32_t* h_status = nullptr;
  size_t h_status_cap = 0;  // words
  hipEvent_t in_done = nullptr, decode_done = nullptr, out_done = nullptr;
  bool busy = false;
  size_t n_chunks = 0;
  uint64_t part_index = 0;
};

template <typename T>
btr_status grow_device(T*& ptr, size_t& cap, size_t need) {
  if (need <= cap) return BTR_OK;
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  cap = 0;
  const size_t want = need + need / 8 + 256;
  hipError_t e = hipMalloc((void**)&ptr, want);
  if (e == hipErrorOutOfMemory) return fail(BTR_ERR_OUT_OF_MEMORY, "hipMalloc: out of device memory");
  if (e != hipSuccess) return hip_fail(e, "hipMalloc");
  cap = want;
  return BTR_OK;
}
template <typename T>
btr_status grow_pinned(T*& ptr, size_t& cap, size_t need_elems) {
  if (need_elems <= cap) return BTR_OK;
  if (ptr) (void)hipHostFree(ptr);
  ptr = nullptr;
  cap = 0;
  const size_t want = need_elems + need_elems / 8 + 64;
  hipError_t e = hipHostMalloc((void**)&ptr, want * sizeof(T), hipHostMallocDefault);
  if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
  cap = want;
  return BTR_OK;
}

size_t round256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

struct btr_gpu_pipeline {
  int device = 0;
  btr_gpu_ctx* ctx = nullptr;
  std::vector<PipeSlot> slots;
  hipStream_t s_in = nullptr, s_dec = nullptr, s_out = nullptr;
  uint64_t submitted = 0;
  uint64_t acquired = 0;  // parts handed out by acquire and not yet submitted: 0 or 1
  uint32_t status_or = 0;
  uint64_t first_bad_part = ~0ull;
  btr_pipeline_stats stats{};
};

namespace {

// Waits until the slot's last part has left the device and folds its status words in.
btr_status retire(btr_gpu_pipeline* p, PipeSlot& s) {
  if (!s.busy) return BTR_OK;
  const double t0 = now_s();
  HIP_TRY(hipEventSynchronize(s.out_done));
  p->stats.wait_seconds += now_s() - t0;
  s.busy = false;
  uint32_t any = 0;
  for (size_t i = 0; i < s.n_chunks; i++) any |= s.h_status[i];
  if (any) {
    p->status_or |= any;
    p->first_bad_part = std::min(p->first_bad_part, s.part_index);
  }
  return BTR_OK;
}

}  // namespace

extern "C" {

btr_status btr_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
  return BTR_OK;
}
btr_status btr_host_free(void* ptr) {
  if (ptr) HIP_TRY(hipHostFree(ptr));
  return BTR_OK;
}

btr_status btr_gpu_pipeline_create(btr_gpu_ctx* ctx, uint32_t slots, btr_gpu_pipeline** out) {
  if (!ctx || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (slots == 0) slots = 3;
  if (slots < 2 || slots > 16) return fail(BTR_ERR_INVALID_ARGUMENT, "a pipeline has 2 to 16 slots");
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  auto p = new btr_gpu_pipeline();
  p->device = ctx->device;
  p->ctx = ctx;
  p->slots.resize(slots);
  auto cleanup = [&](btr_status st) {
    btr_gpu_pipeline_destroy(p);
    return st;
  };
  hipError_t e;
  p->s_in = ctx->pipe[0], p->s_dec = ctx->pipe[1], p->s_out = ctx->pipe[2];  // owned by the context
  for (PipeSlot& s : p->slots)
    for (hipEvent_t* ev : {&s.in_done, &s.decode_done, &s.out_done})
      if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return cleanup(hip_fail(e, "hipEventCreate"));
  *out = p;
  return BTR_OK;
}

void btr_gpu_pipeline_destroy(btr_gpu_pipeline* p) {
  if (!p) return;
  DeviceGuard guard(p->device);
  for (hipStream_t s : {p->s_in, p->s_dec, p->s_out})
    if (s) (void)hipStreamSynchronize(s);
  for (PipeSlot& s : p->slots) {
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.d_nm) (void)hipFree(s.d_nm);
    if (s.d_arena) (void)hipFree(s.d_arena);
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_stage) (void)hipHostFree(s.h_stage);
    if (s.h_status) (void)hipHostFree(s.h_status);
    for (hipEvent_t ev : {s.in_done, s.decode_done, s.out_done})
      if (ev) (void)hipEventDestroy(ev);
  }
  delete p;
}

btr_status btr_gpu_pipeline_acquire(btr_gpu_pipeline* p, size_t bytes, void** host_buffer) {
  if (!p || !host_buffer) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (p->acquired) return fail(BTR_ERR_INVALID_ARGUMENT, "the buffer handed out last has not been submitted");
  DeviceGuard guard(p->device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  PipeSlot& s = p->slots[p->submitted % p->slots.size()];
  btr_status st = retire(p, s);
  if (st != BTR_OK) return st;
  if ((st = grow_pinned(s.h_in, s.h_in_cap, bytes + 16)) != BTR_OK) return st;
  p->acquired = 1;
  *host_buffer = s.h_in;
  return BTR_OK;
}

btr_status btr_gpu_pipeline_submit(btr_gpu_pipeline* p, const btr_part_job* job) {
  if (!p || !job || !job->host_part) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  if (!job->host_dst && !job->dev_dst) return fail(BTR_ERR_INVALID_ARGUMENT, "a job needs a host or a device destination");
  if (job->bytes > 0xFFFFFFFFull) return fail(BTR_ERR_UNSUPPORTED, "part above 4 GiB");
  DeviceGuard guard(p->device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  PipeSlot& s = p->slots[p->submitted % p->slots.size()];
  btr_status st = retire(p, s);
  if (st != BTR_OK) return st;
  p->acquired = 0;

  // ---- host: envelopes, sizes, programs -------------------------------------------------------
  const double t0 = now_s();
  uint32_t count = 0;
  if ((st = btr_part_chunk_count(job->host_part, job->bytes, &count)) != BTR_OK) return st;
  std::vector<btr_chunk_info> infos(count);
  if (count && (st = btr_part_inspect(job->host_part, job->bytes, 0, count, infos.data(), nullptr, nullptr)) != BTR_OK)
    return st;
  size_t out_bytes = 0, rows = 0;
  for (uint32_t k = 0; k < count; k++) {
    out_bytes = (out_bytes + 15) / 16 * 16 + infos[k].decoded_bytes;
    rows += infos[k].tuple_count;
  }
  if (out_bytes > job->dst_bytes) return fail(BTR_ERR_BUFFER_TOO_SMALL, "destination too small for the part");
  const bool want_nm = job->host_nullmap_dst || job->dev_nullmap_dst;
  if (want_nm && rows > job->nullmap_bytes) return fail(BTR_ERR_BUFFER_TOO_SMALL, "nullmap destination too small");
  if ((st = grow_device(s.d_in, s.d_in_cap, job->bytes + 16)) != BTR_OK) return st;
  uint8_t* d_out = (uint8_t*)job->dev_dst;
  if (!d_out) {
    if ((st = grow_device(s.d_out, s.d_out_cap, out_bytes + 16)) != BTR_OK) return st;
    d_out = s.d_out;
  }
  uint8_t* d_nm = nullptr;
  if (want_nm) {
    d_nm = job->dev_nullmap_dst;
    if (!d_nm) {
      if ((st = grow_device(s.d_nm, s.d_nm_cap, rows + 16)) != BTR_OK) return st;
      d_nm = s.d_nm;
    }
  }
  std::vector<btr_chunk_desc> descs;
  if ((st = part_descs(job->host_part, s.d_in, job->bytes, 0, count, d_out, out_bytes + 16, d_nm, rows + 16, descs)) != BTR_OK)
    return st;
  std::vector<WorkItem> items;
  std::vector<BuiltProgram> progs;
  btr_plan_stats stats{};
  Layout lo;
  try {
    if ((st = compile_chunks(descs.data(), count, items, progs, stats)) != BTR_OK) return st;
    lay_out(items, progs, stats, lo);
  } catch (const std::bad_alloc&) {
    return fail(BTR_ERR_OUT_OF_MEMORY, "out of host memory while compiling the plan");
  }
  const size_t order_off = round256(count * sizeof(WorkItem));
  const size_t status_off = order_off + round256((size_t)count * sizeof(LaunchEntry));
  const size_t prog_off = status_off + round256((size_t)count * 4);
  const size_t tables_off = prog_off + round256(lo.blob.size() + 16);
  if ((st = grow_device(s.d_arena, s.d_arena_cap, tables_off + lo.scratch)) != BTR_OK) return st;
  if ((st = grow_pinned(s.h_stage, s.h_stage_cap, tables_off)) != BTR_OK) return st;
  if ((st = grow_pinned(s.h_status, s.h_status_cap, (size_t)count + 1)) != BTR_OK) return st;
  for (size_t i = 0; i < count; i++) {
    items[i].scratch = progs[i].scratch_bytes ? (uint64_t)(uintptr_t)(s.d_arena + tables_off) + items[i].scratch : 0;
    items[i].debug_flags = 0;
    items[i].debug_buf = 0;
  }
  memcpy(s.h_stage, items.data(), count * sizeof(WorkItem));
  memcpy(s.h_stage + order_off, lo.order.data(), (size_t)count * sizeof(LaunchEntry));
  memset(s.h_stage + status_off, 0, (size_t)count * 4);
  memcpy(s.h_stage + prog_off, lo.blob.data(), lo.blob.size());
  p->stats.compile_seconds += now_s() - t0;

  // ---- copy-in ---------------------------------------------------------------------------------
  if (job->bytes) HIP_TRY(hipMemcpyAsync(s.d_in, job->host_part, job->bytes, hipMemcpyHostToDevice, p->s_in));
  HIP_TRY(hipMemcpyAsync(s.d_arena, s.h_stage, prog_off + lo.blob.size(), hipMemcpyHostToDevice, p->s_in));
  HIP_TRY(hipEventRecord(s.in_done, p->s_in));
  // ---- decode ----------------------------------------------------------------------------------
  HIP_TRY(hipStreamWaitEvent(p->s_dec, s.in_done, 0));
  const WorkItem* d_items = reinterpret_cast<const WorkItem*>(s.d_arena);
  const LaunchEntry* d_order = reinterpret_cast<const LaunchEntry*>(s.d_arena + order_off);
  uint32_t* d_status = reinterpret_cast<uint32_t*>(s.d_arena + status_off);
  for (const LaunchGroup& lg : lo.launches)
    HIP_TRY(enqueue_launch(lg, p->s_dec, d_items, d_order, s.d_arena + prog_off, d_status));
  HIP_TRY(hipEventRecord(s.decode_done, p->s_dec));
  // ---- copy-out --------------------------------------------------------------------------------
  HIP_TRY(hipStreamWaitEvent(p->s_out, s.decode_done, 0));
  if (job->host_dst && out_bytes) HIP_TRY(hipMemcpyAsync(job->host_dst, d_out, out_bytes, hipMemcpyDeviceToHost, p->s_out));
  if (job->host_nullmap_dst && rows) HIP_TRY(hipMemcpyAsync(job->host_nullmap_dst, d_nm, rows, hipMemcpyDeviceToHost, p->s_out));
  if (count) HIP_TRY(hipMemcpyAsync(s.h_status, d_status, (size_t)count * 4, hipMemcpyDeviceToHost, p->s_out));
  HIP_TRY(hipEventRecord(s.out_done, p->s_out));
  s.busy = true;
  s.n_chunks = count;
  s.part_index = p->submitted++;
  p->stats.parts++;
  p->stats.chunks += count;
  p->stats.stored_bytes += job->bytes;
  p->stats.decoded_bytes += out_bytes;
  p->stats.tuples += rows;
  return BTR_OK;
}

btr_status btr_gpu_pipeline_drain(btr_gpu_pipeline* p, uint32_t* status_or, uint64_t* first_bad_part) {
  if (!p) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGuard guard(p->device);
  if (!guard.ok) return fail(BTR_ERR_DEVICE, "hipSetDevice failed");
  for (PipeSlot& s : p->slots) {
    const btr_status st = retire(p, s);
    if (st != BTR_OK) return st;
  }
  const uint32_t any = p->status_or;
  const uint64_t first = p->first_bad_part;
  if (status_or) *status_or = any;
  if (first_bad_part) *first_bad_part = first;
  p->status_or = 0;
  p->first_bad_part = ~0ull;
  if (any) return fail(BTR_ERR_CORRUPT, "malformed compressed stream in part " + std::to_string(first));
  return BTR_OK;
}

btr_status btr_gpu_pipeline_stats(const btr_gpu_pipeline* p, btr_pipeline_stats* out) {
  if (!p || !out) return fail(BTR_ERR_INVALID_ARGUMENT, "null argument");
  *out = p->stats;
  return BTR_OK;
}

}  // extern "C"
