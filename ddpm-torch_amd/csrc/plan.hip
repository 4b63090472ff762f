// Launch-plan executor: the training step's ~490 C-ABI calls recorded once, re-issued by ONE call per plan segment.
//
// Why: `Trainer.step` (ddpm_torch/utils/train.py:148-170 upstream) is a fixed, shape-static sequence of launches; issuing it from
// Python costs 6-7 ms of interpreter + ctypes time per ~9.4 ms GPU step, and a replayed hipGraph serialises the two streams the step
// uses (main chain | weight-gradient leaves).  A plan keeps EAGER semantics — every entry is the same `ddpm_*` call with the same
// arguments on the same hipStream_t, the side stream stays concurrent, RCCL calls run between segments — and takes the interpreter out.
//
// An entry = (thunk of an exported `int ddpm_xxx(...)`, its arguments as 64-bit words).  The thunks are generated from the declarations
// in include/ddpm_hip.h by a variadic template, so an entry point whose signature changes cannot be called with a stale layout: the
// argument count is checked when the entry is appended.  Nothing here allocates device memory or synchronises.
#include "common.h"
#include "../../include/ddpm_hip.h"
#include "../../include/ddpm_hip_debug.h"

#include <cstring>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

typedef unsigned long long word_t;

template <class T> inline T from_word(word_t w) {
    if constexpr (std::is_pointer<T>::value) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(w));
    } else if constexpr (std::is_same<T, float>::value) {
        const unsigned u = static_cast<unsigned>(w);
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    } else {
        return static_cast<T>(w);            // int (two's complement in the low 32 bits), long long, unsigned long long
    }
}

typedef int (*thunk_t)(const word_t*);

template <class... A, size_t... I> inline int invoke(int (*fn)(A...), const word_t* w, std::index_sequence<I...>) {
    return fn(from_word<A>(w[I])...);
}
template <class... A> constexpr int arity(int (*)(A...)) { return (int)sizeof...(A); }
template <class... A> inline int invoke_all(int (*fn)(A...), const word_t* w) { return invoke(fn, w, std::index_sequence_for<A...>{}); }

struct EntryPoint {
    const char* name;
    thunk_t thunk;
    int nargs;
};

#define EP(fn) {#fn, [](const word_t* w) -> int { return invoke_all(&fn, w); }, arity(&fn)}

}  // namespace

extern "C" int ddpm_stream_order(void* waiter, void* signaller);
extern "C" int ddpm_fill_zero(void* p, long long bytes, void* stream);

namespace {

// every launching entry point of include/ddpm_hip.h (the pure queries — *_variant, *_splits, ddpm_gn_workspace_floats — enqueue nothing)
const EntryPoint kEntryPoints[] = {
    EP(ddpm_conv2d_nhwc), EP(ddpm_conv2d_wgrad_nhwc), EP(ddpm_conv3x3_wgrad_nhwc), EP(ddpm_conv3x3_wgrad_up_nhwc), EP(ddpm_conv1x1_wgrad_nhwc),
    EP(ddpm_wgrad_reduce), EP(ddpm_wgrad_unpack), EP(ddpm_wgrad_unpack_sumsq), EP(ddpm_gemm),
    EP(ddpm_attention_fwd), EP(ddpm_attention_fwd_lse), EP(ddpm_attention_bwd), EP(ddpm_groupnorm_silu_fwd),
    EP(ddpm_groupnorm_silu_bwd), EP(ddpm_timestep_embedding), EP(ddpm_nchw_to_nhwc), EP(ddpm_pack_weight), EP(ddpm_pack_weight_multi),
    EP(ddpm_q_sample), EP(ddpm_mse_fwd), EP(ddpm_mse_bwd), EP(ddpm_weighted_sum_f32), EP(ddpm_atb_f32), EP(ddpm_p_sample_step), EP(ddpm_vlb_terms), EP(ddpm_vlb_terms_bwd), EP(ddpm_gather_i64),
    EP(ddpm_add_i64), EP(ddpm_gather_rows_f32), EP(ddpm_silu_fwd), EP(ddpm_silu_bwd), EP(ddpm_colsum), EP(ddpm_upsample2x_bwd), EP(ddpm_resample2x_nhwc),
    EP(ddpm_add_rows), EP(ddpm_softmax_fwd), EP(ddpm_softmax_bwd), EP(ddpm_sumsq_accumulate), EP(ddpm_adam_ema_step), EP(ddpm_mt_grad_sumsq),
    EP(ddpm_mt_adam_ema), EP(ddpm_mt_gather_f32), EP(ddpm_mfma_probe), EP(ddpm_copy_probe), EP(ddpm_dropout_mask), EP(ddpm_stream_order), EP(ddpm_fill_zero),
};
constexpr int kNumEntryPoints = (int)(sizeof(kEntryPoints) / sizeof(kEntryPoints[0]));

struct Entry {
    thunk_t thunk;
    int first_word;       // index into Plan::words
    int ep;               // index into kEntryPoints (for diagnostics)
};

struct Plan {
    std::vector<Entry> entries;
    std::vector<word_t> words;
    std::vector<int> segment_end;     // entries [segment_end[s-1], segment_end[s]) form segment s; the open segment ends at entries.size()
    int failed_index = -1;
    int failed_status = 0;
};

// ---- events for ddpm_stream_order: a per-device ring, created on first use, never destroyed (process lifetime).  Re-recording an event
// that an earlier hipStreamWaitEvent still refers to is well defined: a wait binds to the record that precedes it in program order.
constexpr int kRing = 2048;
struct EventRing {
    hipEvent_t ev[kRing];
    unsigned next = 0;
    bool ready = false;
};
EventRing g_rings[64];
std::mutex g_ring_mutex;

}  // namespace

extern "C" {

/* `waiter` (a hipStream_t) will not run anything enqueued after this call before everything enqueued on `signaller` so far has finished:
 * hipEventRecord + hipStreamWaitEvent on a pooled event.  The fork / join edges between the step's main stream and its weight-gradient
 * stream (torch.cuda.Event record / wait in earlier rounds), as one plan-recordable C call. */
int ddpm_stream_order(void* waiter, void* signaller) {
    if (waiter == signaller) return DDPM_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DDPM_ERR_LAUNCH;
    EventRing& r = g_rings[dev & 63];
    hipEvent_t e;
    {
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        if (!r.ready) {
            for (int i = 0; i < kRing; ++i)
                if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) return DDPM_ERR_LAUNCH;
            r.ready = true;
        }
        e = r.ev[r.next++ % kRing];
    }
    if (hipEventRecord(e, (hipStream_t)signaller) != hipSuccess) return DDPM_ERR_LAUNCH;
    if (hipStreamWaitEvent((hipStream_t)waiter, e, 0) != hipSuccess) return DDPM_ERR_LAUNCH;
    return DDPM_OK;
}

/* bytes of zero at p, stream-ordered (torch.zeros / Tensor.zero_ of the gradient staging buffers, as a plan-recordable call) */
int ddpm_fill_zero(void* p, long long bytes, void* stream) {
    if (bytes < 0) return DDPM_ERR_SHAPE;
    if (bytes == 0) return DDPM_OK;
    if (!p) return DDPM_ERR_NULL;
    return hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream) == hipSuccess ? DDPM_OK : DDPM_ERR_LAUNCH;
}

void* ddpm_plan_create(void) { return new Plan(); }

int ddpm_plan_destroy(void* plan) {
    delete static_cast<Plan*>(plan);
    return DDPM_OK;
}

/* Append one call.  `entry` = exported name; `words` = its arguments in declaration order, one 64-bit word each (pointers and 64-bit
 * integers as they are, int sign- or zero-extended, float as its IEEE-754 bits in the low half).  Returns the entry's index, or
 * -DDPM_ERR_SHAPE for an unknown name / wrong argument count, -DDPM_ERR_NULL for null arguments. */
int ddpm_plan_append(void* plan, const char* entry, const unsigned long long* words, int n_words) {
    if (!plan || !entry || (!words && n_words > 0)) return -DDPM_ERR_NULL;
    Plan* p = static_cast<Plan*>(plan);
    for (int i = 0; i < kNumEntryPoints; ++i) {
        if (std::strcmp(kEntryPoints[i].name, entry) != 0) continue;
        if (kEntryPoints[i].nargs != n_words) return -DDPM_ERR_SHAPE;
        Entry e;
        e.thunk = kEntryPoints[i].thunk;
        e.first_word = (int)p->words.size();
        e.ep = i;
        p->words.insert(p->words.end(), words, words + n_words);
        p->entries.push_back(e);
        return (int)p->entries.size() - 1;
    }
    return -DDPM_ERR_SHAPE;
}

/* Close the segment being built; returns its index.  The host may do anything between two segments of a run (the data-parallel
 * step issues its RCCL all-reduces there). */
int ddpm_plan_cut(void* plan) {
    if (!plan) return -DDPM_ERR_NULL;
    Plan* p = static_cast<Plan*>(plan);
    p->segment_end.push_back((int)p->entries.size());
    return (int)p->segment_end.size() - 1;
}

int ddpm_plan_segments(void* plan) {
    if (!plan) return -DDPM_ERR_NULL;
    Plan* p = static_cast<Plan*>(plan);
    const int closed = (int)p->segment_end.size();
    const int last_end = closed ? p->segment_end.back() : 0;
    return closed + ((int)p->entries.size() > last_end ? 1 : 0);
}

int ddpm_plan_entries(void* plan) { return plan ? (int)static_cast<Plan*>(plan)->entries.size() : -DDPM_ERR_NULL; }

/* Issue every call of one segment, in recorded order, each on the stream it was recorded with.  Stops at the first call that fails and
 * returns its status code (ddpm_plan_failed_entry then names it); 0 = everything enqueued. */
int ddpm_plan_run(void* plan, int segment) {
    if (!plan) return DDPM_ERR_NULL;
    Plan* p = static_cast<Plan*>(plan);
    const int closed = (int)p->segment_end.size();
    if (segment < 0 || segment > closed) return DDPM_ERR_SHAPE;
    const int first = segment ? p->segment_end[segment - 1] : 0;
    const int last = segment < closed ? p->segment_end[segment] : (int)p->entries.size();
    const word_t* words = p->words.data();
    const Entry* e = p->entries.data();
    for (int i = first; i < last; ++i) {
        const int rc = e[i].thunk(words + e[i].first_word);
        if (rc != DDPM_OK) {
            p->failed_index = i;
            p->failed_status = rc;
            return rc;
        }
    }
    return DDPM_OK;
}

/* name of the entry point the last failing ddpm_plan_run stopped at (NULL: none failed); *index receives its position in the plan */
const char* ddpm_plan_failed_entry(void* plan, int* index) {
    if (!plan) return nullptr;
    Plan* p = static_cast<Plan*>(plan);
    if (index) *index = p->failed_index;
    return p->failed_index < 0 ? nullptr : kEntryPoints[p->entries[p->failed_index].ep].name;
}

/* does the executor know this entry point, and with how many arguments?  (-1: unknown) */
int ddpm_plan_entry_arity(const char* entry) {
    if (!entry) return -1;
    for (int i = 0; i < kNumEntryPoints; ++i)
        if (std::strcmp(kEntryPoints[i].name, entry) == 0) return kEntryPoints[i].nargs;
    return -1;
}

}  // extern "C"
