// Launch tape: record the kernel launches of a training step once, re-issue them with one C call per segment (dn_tape_* in
// dispnet_hip.h; the DN_LAUNCH wrapper in dn_internal.h does the recording).
//
// Why not a hipGraph: measured on ROCm 7.2 the replay of the captured step is SLOWER on the device than the same launches issued
// eagerly (4 images: 5.43 vs 5.19 ms, profiles/r03_a_strong_1gpu.txt) -- a pre-queued eager sequence runs at the device's floor, the
// graph does not.  The tape re-issues the plain launches (same streams, same event fences between the two compute streams), so the
// device sees exactly the eager sequence while the host spends ~2 us per launch instead of ~20.
//
// A tape holds, in issue order: kernel launches (closure over kernel, grid, block, LDS bytes, stream, by-value arguments), stream
// fences (record an event on one stream, make another wait for it) and marks.  Marks cut the tape into segments: the caller replays
// segment by segment and does its own host work in between (a gradient bucket's all-reduce on another library's communicator).
#include <chrono>
#include <mutex>
#include <vector>

#include "dn_internal.h"

namespace dn {

struct TapeOp {
  std::function<void(hipEvent_t)> launch;   // a kernel launch (called with `stop`), or empty for a fence
  const void* what = nullptr;               // the kernel's host function (dn_tape_replay_timed names it), nullptr for a fence
  hipStream_t stream = nullptr;             // launch: its stream
  hipEvent_t stop = nullptr;                // launch: the event a fence behind it rides on (its stop event), or nullptr
  hipStream_t waiter = nullptr, waitee = nullptr;   // fence
  hipEvent_t ev = nullptr;                  // fence: its event
  bool record = true;                       // fence: record `ev` on `waitee` at replay (false: `ev` is the stop event of an earlier launch)
  void run() const {
    if (launch) {
      launch(stop);
    } else {
      if (record) (void)hipEventRecord(ev, waitee);
      (void)hipStreamWaitEvent(waiter, ev, 0);
    }
  }
};

struct LaunchTape {
  std::vector<TapeOp> ops;
  std::vector<size_t> marks;           // ops.size() at each dn_tape_mark
  std::vector<hipEvent_t> events;
  std::mutex mu;                       // forward is recorded on the caller's thread, backward on autograd's
  size_t launches = 0, fences = 0, riding = 0;
  size_t no_ride_before = 0;         // ops in front of the last dn_tape_pause(.., 1): unrecorded work may sit behind them on their streams
};

std::atomic<LaunchTape*> g_tape_rec{nullptr};

void tape_push(LaunchTape* t, std::function<void(hipEvent_t)>&& op, const void* kernel, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(t->mu);
  TapeOp o;
  o.launch = std::move(op);
  o.what = kernel;
  o.stream = stream;
  t->ops.emplace_back(std::move(o));
  ++t->launches;
}

}  // namespace dn

extern "C" {

void* dn_tape_begin(void) {
  dn::LaunchTape* t = new dn::LaunchTape();
  dn::LaunchTape* expected = nullptr;
  if (!dn::g_tape_rec.compare_exchange_strong(expected, t)) {
    delete t;
    dn::set_error("dn_tape_begin: another tape is being recorded");
    return nullptr;
  }
  return t;
}

int dn_tape_end(void* tape) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  dn::LaunchTape* expected = t;
  if (t == nullptr || !dn::g_tape_rec.compare_exchange_strong(expected, nullptr)) {
    dn::set_error("dn_tape_end: this tape is not the one being recorded");
    return DN_ERR_BAD_ARG;
  }
  return DN_OK;
}

int dn_tape_pause(void* tape, int32_t paused) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  if (t == nullptr) {
    dn::set_error("dn_tape_pause: null tape");
    return DN_ERR_BAD_ARG;
  }
  dn::LaunchTape* expected = paused ? t : nullptr;
  if (!dn::g_tape_rec.compare_exchange_strong(expected, paused ? nullptr : t)) {
    dn::set_error("dn_tape_pause: the tape is not in the state this call expects");
    return DN_ERR_BAD_ARG;
  }
  if (paused) {
    // Work enqueued while the tape is paused is live at replay too but invisible here: a later device-scope fence must not ride on a
    // launch recorded BEFORE the pause (its stop event would not cover the unrecorded work behind it)
    std::lock_guard<std::mutex> lock(t->mu);
    t->no_ride_before = t->ops.size();
  }
  return DN_OK;
}

static int tape_fence(void* tape, dn_stream_t waiter, dn_stream_t waitee, unsigned flags, bool ride) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  if (t == nullptr || dn::g_tape_rec.load() != t) {
    dn::set_error("dn_tape_fence: this tape is not being recorded");
    return DN_ERR_BAD_ARG;
  }
  hipEvent_t ev;
  hipError_t e = hipEventCreateWithFlags(&ev, flags);
  if (e != hipSuccess) {
    dn::set_error("dn_tape_fence: hipEventCreate: %s", hipGetErrorString(e));
    return DN_ERR_LAUNCH;
  }
  hipStream_t w = dn::as_stream(waiter), s = dn::as_stream(waitee);
  std::lock_guard<std::mutex> lock(t->mu);
  t->events.push_back(ev);
  dn::TapeOp f;
  f.waiter = w;
  f.waitee = s;
  f.ev = ev;
  // A device-scope fence whose waitee's LAST recorded item is a kernel launch rides on that launch: the event becomes the launch's stop
  // event (the dispatch packet's own completion signal) instead of a marker packet behind it -- tools/ubench/fence_cost.hip: 1.5-4.5 us of
  // the waitee's queue per fence instead of 2.3-6.6.  "Last item": no later launch on that stream and no later fence that made it wait
  // (an in-order stream: the launch's completion then covers everything enqueued on it so far).
  if (ride && !dn::knobs().no_riding_fences) {
    for (size_t i = t->ops.size(); i-- > 0;) {
      dn::TapeOp& o = t->ops[i];
      if (o.launch) {
        if (o.stream != s) continue;
        if ((t->marks.empty() || i >= t->marks.back()) && i >= t->no_ride_before) {      // (within the current segment, not across a pause)
          if (o.stop == nullptr) o.stop = ev;
          f.ev = o.stop;                                     // (a second fence behind the same launch shares its stop event)
          f.record = false;
          ++t->riding;
        }
        break;
      }
      if (o.waiter == s || o.waitee == s) break;      // the stream's last item is a wait / a marker, not a launch
    }
  }
  t->ops.emplace_back(std::move(f));
  ++t->fences;
  return DN_OK;
}

int dn_tape_fence(void* tape, dn_stream_t waiter, dn_stream_t waitee) { return tape_fence(tape, waiter, waitee, hipEventDisableTiming, false); }

// Both streams' work stays on this device: no system-scope release at the event (tools/ubench/fence_cost.hip: the recording stream's queue
// loses 2.3-5.9 us per fence instead of 4.5-8.5; a 4-image step records ~60 of them on its critical stream).
int dn_tape_fence_device(void* tape, dn_stream_t waiter, dn_stream_t waitee) {
  return tape_fence(tape, waiter, waitee, hipEventDisableTiming | hipEventDisableSystemFence, true);
}

int32_t dn_tape_mark(void* tape) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  if (t == nullptr) {
    dn::set_error("dn_tape_mark: null tape");
    return -1;
  }
  std::lock_guard<std::mutex> lock(t->mu);
  t->marks.push_back(t->ops.size());
  return (int32_t)t->marks.size();      // number of the segment that starts here
}

int32_t dn_tape_segments(void* tape) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  return t == nullptr ? -1 : (int32_t)t->marks.size() + 1;
}

int64_t dn_tape_launches(void* tape) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  return t == nullptr ? -1 : (int64_t)t->launches;
}

int64_t dn_tape_riding_fences(void* tape) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  return t == nullptr ? -1 : (int64_t)t->riding;
}

int64_t dn_tape_fences(void* tape) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  return t == nullptr ? -1 : (int64_t)t->fences;
}

int dn_tape_replay(void* tape, int32_t segment) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  if (t == nullptr || dn::g_tape_rec.load() == t) {
    dn::set_error("dn_tape_replay: null tape, or the tape is still being recorded");
    return DN_ERR_BAD_ARG;
  }
  const int32_t nseg = (int32_t)t->marks.size() + 1;
  if (segment < -1 || segment >= nseg) {
    dn::set_error("dn_tape_replay: segment %d of %d", segment, nseg);
    return DN_ERR_BAD_ARG;
  }
  const size_t lo = segment <= 0 ? 0 : t->marks[segment - 1];
  const size_t hi = (segment == -1 || segment == nseg - 1) ? t->ops.size() : t->marks[segment];
  for (size_t i = lo; i < hi; ++i) t->ops[i].run();
  return dn::check_launch("dn_tape_replay");
}

// Diagnostics: replay the whole tape and report the HOST time of every op (ns[i]) and what it is (name[i]: the kernel's name, "fence" for
// a fence; pointers valid while the library is loaded).  Returns the number of ops (<= cap are reported).  A launch that takes the host
// tens of microseconds is one the runtime blocked in (tools/tape_host_profile.py).
int32_t dn_tape_replay_timed(void* tape, int64_t* ns, const char** name, int32_t cap) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  if (t == nullptr || dn::g_tape_rec.load() == t) {
    dn::set_error("dn_tape_replay_timed: null tape, or the tape is still being recorded");
    return -1;
  }
  const size_t n = t->ops.size();
  for (size_t i = 0; i < n; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    t->ops[i].run();
    const auto t1 = std::chrono::steady_clock::now();
    if ((int32_t)i < cap) {
      ns[i] = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
      name[i] = t->ops[i].what != nullptr ? hipKernelNameRefByPtr(t->ops[i].what, nullptr) : (t->ops[i].record ? "fence" : "fence (riding)");
    }
  }
  if (dn::check_launch("dn_tape_replay_timed") != DN_OK) return -1;
  return (int32_t)n;
}

void dn_tape_free(void* tape) {
  dn::LaunchTape* t = reinterpret_cast<dn::LaunchTape*>(tape);
  if (t == nullptr) return;
  dn::LaunchTape* expected = t;
  dn::g_tape_rec.compare_exchange_strong(expected, nullptr);
  for (hipEvent_t ev : t->events) (void)hipEventDestroy(ev);
  delete t;
}

}  // extern "C"
