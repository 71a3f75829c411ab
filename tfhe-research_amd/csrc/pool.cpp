// pool.cpp -- multi-GPU behind the C ABI: one tfhe_pool = one tfhe_context per listed device.
//
// What it stands for in the reference: ONE BootstrappingKey (bootstrapping.rs:18-21) and many independent calls of
// bootstrap() (bootstrapping.rs:58-65) / the boolean gates (boolean.rs:9-53).  Bootstraps do not talk to each other,
// so a batch is cut into contiguous slices, one per device (the same rule as sharding.shard_range of the
// torch.distributed path: the first batch % n members get one more), the read-only keys are replicated, and there is
// no collective anywhere in the data path.
//
//   keys     uploaded and transformed ONCE (member 0: one H2D copy of the raw u32 key, one bsk_prepare launch), then
//            the PREPARED key and the key-switching key go device to device to every other member -- peer copies over
//            xGMI, issued on the destination members' streams so that they travel over different links at the same
//            time; not N host uploads and N prepares.
//   batches  host-pointer calls run one host thread per member (a blocking H2D copy, the kernels and the D2H copy of
//            one member overlap with those of the others; every member has its own stream); the `_device` form takes
//            one device pointer per member and only enqueues.
// A device may be listed more than once (two members on one GPU): that is how the pool is tested on a one-GPU box.
#include "context.h"

#include <cstring>
#include <new>
#include <thread>

struct tfhe_pool {
  std::vector<tfhe_context*> members;
  std::string last_error;
};

namespace {

int pool_fail(tfhe_pool* pool, int status, const std::string& msg) {
  if (pool) pool->last_error = msg;
  return status;
}

int member_fail(tfhe_pool* pool, size_t i, int status) {
  return pool_fail(pool, status, "member " + std::to_string(i) + " (device " + std::to_string(pool->members[i]->device) +
                                     "): " + tfhe_last_error(pool->members[i]));
}

void shard(size_t batch, size_t n, size_t i, size_t* first, size_t* count) {
  const size_t base = batch / n, extra = batch % n;
  *first = i * base + (i < extra ? i : extra);
  *count = base + (i < extra ? 1 : 0);
}

// run fn(member index) for every member with work on its own host thread; the first failing member's status wins.
// Nothing may propagate out of an extern "C" entry point: if a thread cannot be started (EAGAIN, bad_alloc) that
// member's slice runs inline on the calling thread instead, and whatever was started is joined before returning.
template <class Fn>
int for_members(tfhe_pool* pool, size_t batch, Fn fn) {
  const size_t n = pool->members.size();
  std::vector<int> status(n, TFHE_OK);
  std::vector<std::thread> threads;
  try {
    threads.reserve(n);
  } catch (...) {
    return pool_fail(pool, TFHE_ERR_HIP, "out of host memory");
  }
  for (size_t i = 0; i < n; ++i) {
    size_t first, count;
    shard(batch, n, i, &first, &count);
    if (count == 0) continue;
    bool started = false;
    if (n > 1) {
      try {
        threads.emplace_back([&status, &fn, i, first, count] { status[i] = fn(i, first, count); });
        started = true;
      } catch (...) {
        started = false;  // no thread to be had: this member's slice runs here
      }
    }
    if (!started) status[i] = fn(i, first, count);
  }
  for (auto& t : threads) t.join();
  for (size_t i = 0; i < n; ++i)
    if (status[i] != TFHE_OK) return member_fail(pool, i, status[i]);
  return TFHE_OK;
}

}  // namespace

extern "C" {

int tfhe_pool_create(const tfhe_params* params, const int* devices, size_t n_devices, int backend, tfhe_pool** out) {
  if (!params || !devices || !out || n_devices == 0 || n_devices > 64) return TFHE_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  tfhe_pool* pool = new (std::nothrow) tfhe_pool();
  if (!pool) return TFHE_ERR_HIP;
  for (size_t i = 0; i < n_devices; ++i) {
    tfhe_context* ctx = nullptr;
    int st = tfhe_context_create_with_backend(params, devices[i], backend, &ctx);
    if (st != TFHE_OK) {
      tfhe_pool_destroy(pool);
      return st;
    }
    pool->members.push_back(ctx);
  }
  *out = pool;
  return TFHE_OK;
}

void tfhe_pool_destroy(tfhe_pool* pool) {
  if (!pool) return;
  for (tfhe_context* ctx : pool->members) tfhe_context_destroy(ctx);
  delete pool;
}

size_t tfhe_pool_size(const tfhe_pool* pool) { return pool ? pool->members.size() : 0; }

tfhe_context* tfhe_pool_member(tfhe_pool* pool, size_t i) {
  return pool && i < pool->members.size() ? pool->members[i] : nullptr;
}

const char* tfhe_pool_last_error(const tfhe_pool* pool) { return pool ? pool->last_error.c_str() : ""; }

int tfhe_pool_shard(const tfhe_pool* pool, size_t batch, size_t member, size_t* first, size_t* count) {
  if (!pool || !first || !count || member >= pool->members.size()) return TFHE_ERR_INVALID_ARGUMENT;
  shard(batch, pool->members.size(), member, first, count);
  return TFHE_OK;
}

int tfhe_pool_set_decomposer_alignment(tfhe_pool* pool, int aligned) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < pool->members.size(); ++i) {
    int st = tfhe_context_set_decomposer_alignment(pool->members[i], aligned);
    if (st) return member_fail(pool, i, st);
  }
  return TFHE_OK;
}

int tfhe_pool_set_kernel_shape(tfhe_pool* pool, int shape) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < pool->members.size(); ++i) {
    int st = tfhe_context_set_kernel_shape(pool->members[i], shape);
    if (st) return member_fail(pool, i, st);
  }
  return TFHE_OK;
}

int tfhe_pool_set_bootstrap_order(tfhe_pool* pool, int ks_first) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < pool->members.size(); ++i) {
    int st = tfhe_context_set_bootstrap_order(pool->members[i], ks_first);
    if (st) return member_fail(pool, i, st);
  }
  return TFHE_OK;
}

int tfhe_pool_reserve(tfhe_pool* pool, size_t max_batch) {
  if (!pool || max_batch == 0) return TFHE_ERR_INVALID_ARGUMENT;
  const size_t n = pool->members.size();
  for (size_t i = 0; i < n; ++i) {
    size_t first, count;
    shard(max_batch, n, i, &first, &count);
    if (count == 0) continue;
    int st = tfhe_context_reserve(pool->members[i], count);
    if (st) return member_fail(pool, i, st);
  }
  return TFHE_OK;
}

int tfhe_pool_synchronize(tfhe_pool* pool) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < pool->members.size(); ++i) {
    int st = tfhe_context_synchronize(pool->members[i]);
    if (st) return member_fail(pool, i, st);
  }
  return TFHE_OK;
}

// prepare once on member 0, replicate the prepared key: see the header of this file
static int replicate_from_member0(tfhe_pool* pool) {
  // no member keeps an older key while the new one travels: if a copy fails, the members it did not reach answer
  // TFHE_ERR_NO_KEY instead of bootstrapping their slices under the key loaded before
  for (size_t i = 1; i < pool->members.size(); ++i) pool->members[i]->have_key = false;
  for (size_t i = 1; i < pool->members.size(); ++i) {
    int st = tfhe::host::adopt_prepared_key(pool->members[i], pool->members[0]);
    if (st) return member_fail(pool, i, st);
  }
  return tfhe_pool_synchronize(pool);  // the copies run concurrently on the members' own streams until here
}

int tfhe_pool_replicate_key(tfhe_pool* pool) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  if (!pool->members[0]->have_key) return pool_fail(pool, TFHE_ERR_NO_KEY, "member 0 holds no key to replicate");
  return replicate_from_member0(pool);
}

// bootstrapping_key_gen (bootstrapping.rs:23-56) for the whole pool: generated on member 0's device, and with `load`
// installed on EVERY member (a key installed on member 0 alone would leave the other members without a key -- or, worse,
// bootstrapping their slices under the key loaded before)
static int pool_key_gen(tfhe_pool* pool, const uint32_t* lwe_sk, const uint32_t* glwe_sk, uint32_t* bsk, uint32_t* ksk,
                        int load, bool bmmp) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  int st = (bmmp ? tfhe_bootstrapping_key_gen_bmmp : tfhe_bootstrapping_key_gen)(pool->members[0], lwe_sk, glwe_sk, bsk, ksk, load);
  if (st) return member_fail(pool, 0, st);
  return load ? replicate_from_member0(pool) : TFHE_OK;
}

int tfhe_pool_bootstrapping_key_gen(tfhe_pool* pool, const uint32_t* lwe_sk, const uint32_t* glwe_sk, uint32_t* bsk,
                                    uint32_t* ksk, int load) {
  return pool_key_gen(pool, lwe_sk, glwe_sk, bsk, ksk, load, false);
}

int tfhe_pool_bootstrapping_key_gen_bmmp(tfhe_pool* pool, const uint32_t* lwe_sk, const uint32_t* glwe_sk,
                                         uint32_t* bsk_bmmp, uint32_t* ksk, int load) {
  return pool_key_gen(pool, lwe_sk, glwe_sk, bsk_bmmp, ksk, load, true);
}

int tfhe_pool_load_bootstrapping_key(tfhe_pool* pool, const uint32_t* bsk, const uint32_t* ksk) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  int st = tfhe_load_bootstrapping_key(pool->members[0], bsk, ksk);
  if (st) return member_fail(pool, 0, st);
  return replicate_from_member0(pool);
}

int tfhe_pool_load_bootstrapping_key_bmmp(tfhe_pool* pool, const uint32_t* bsk_bmmp, const uint32_t* ksk) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  int st = tfhe_load_bootstrapping_key_bmmp(pool->members[0], bsk_bmmp, ksk);
  if (st) return member_fail(pool, 0, st);
  return replicate_from_member0(pool);
}

int tfhe_pool_load_bootstrapping_key_device(tfhe_pool* pool, const uint32_t* bsk, const uint32_t* ksk) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  int st = tfhe_load_bootstrapping_key_device(pool->members[0], bsk, ksk);  // pointers on member 0's device
  if (st) return member_fail(pool, 0, st);
  return replicate_from_member0(pool);
}

int tfhe_pool_bootstrap_batch(tfhe_pool* pool, const uint32_t* lwe_in, size_t batch, const uint32_t* tv,
                              size_t tv_count, uint32_t* lwe_out) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  if (!lwe_in || !tv || !lwe_out) return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  if (batch == 0) return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "empty batch");
  if (tv_count != 1 && tv_count != batch) return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "tv_count must be 1 or batch");
  const size_t w = tfhe::host::io_words(pool->members[0]);
  const size_t N = pool->members[0]->N;
  return for_members(pool, batch, [&](size_t i, size_t first, size_t count) {
    return tfhe_bootstrap_batch(pool->members[i], lwe_in + first * w, count, tv_count == 1 ? tv : tv + first * N,
                                tv_count == 1 ? 1 : count, lwe_out + first * w);
  });
}

int tfhe_pool_gate_batch(tfhe_pool* pool, const uint32_t truth[4], const uint32_t* ct0, const uint32_t* ct1,
                         size_t batch, uint32_t* lwe_out) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  if (!truth || !ct0 || !ct1 || !lwe_out) return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  if (batch == 0) return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "empty batch");
  const size_t w = tfhe::host::io_words(pool->members[0]);
  return for_members(pool, batch, [&](size_t i, size_t first, size_t count) {
    return tfhe_gate_batch(pool->members[i], truth, ct0 + first * w, ct1 + first * w, count, lwe_out + first * w);
  });
}

int tfhe_pool_bootstrap_shards_device(tfhe_pool* pool, const uint32_t* const* lwe_in, const size_t* counts,
                                      const uint32_t* const* tv, const size_t* tv_counts, uint32_t* const* lwe_out) {
  if (!pool) return TFHE_ERR_INVALID_ARGUMENT;
  if (!lwe_in || !counts || !tv || !tv_counts || !lwe_out) return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  // validate every member's arguments BEFORE anything is enqueued: a call either launches on all members with work
  // or on none (a failure half way would leave a partially launched step the caller cannot tell from a failed one)
  for (size_t i = 0; i < pool->members.size(); ++i) {
    if (counts[i] == 0) continue;
    tfhe_context* m = pool->members[i];
    if (!m->have_key) return pool_fail(pool, TFHE_ERR_NO_KEY, "member " + std::to_string(i) + " holds no key");
    if (!lwe_in[i] || !tv[i] || !lwe_out[i]) return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "member " + std::to_string(i) + ": null pointer");
    if (tv_counts[i] != 1 && tv_counts[i] != counts[i])
      return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "member " + std::to_string(i) + ": tv_count must be 1 or the shard's count");
    if (counts[i] > 0x7FFFFFFFull) return pool_fail(pool, TFHE_ERR_INVALID_ARGUMENT, "member " + std::to_string(i) + ": shard exceeds 2^31 - 1");
  }
  // enqueue only: one launch sequence per member on its own stream, nothing waits here
  for (size_t i = 0; i < pool->members.size(); ++i) {
    if (counts[i] == 0) continue;
    int st = tfhe_bootstrap_batch_device(pool->members[i], lwe_in[i], counts[i], tv[i], tv_counts[i], lwe_out[i]);
    if (st) return member_fail(pool, i, st);
  }
  return TFHE_OK;
}

}  // extern "C"
