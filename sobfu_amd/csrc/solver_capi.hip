// Solver handle: workspace + the gradient-descent loop of sobfu::device::estimate_psi
// (reference: src/sobfu/cuda/solver.cu:85-205, src/sobfu/solver.cpp:7-101,160-262).
//
// One iteration = pass A + pass B (solver_kernels.hip) on ONE stream.  The reference copies block partials to
// the host and tests convergence on the CPU every iteration (src/sobfu/reductor.cpp:52-57); here the max update
// norm of iteration k lives in 256 device slots, iteration k+1's kernels test it themselves and turn into no-ops
// once it is <= max_update_norm, and the host only looks every kCheckEvery iterations -- the iteration at which
// the solver stops, and every array it leaves behind, are exactly the reference's.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sobfu_device.hpp"
#include "sobfu_hip.h"
#include "sobfu_host.hpp"
#include "sobfu_launch.hpp"


namespace {

constexpr int kSlots      = 256;
constexpr int kCheckEvery = 32;
constexpr int kMaxRunAhead = 96;  // iterations the host may enqueue beyond the last max-norm row it has examined

float host_sqrt_rd(float s) {  // __fsqrt_rd
    float r = std::sqrt(s);
    if (r > 0.f && (double) r * (double) r > (double) s) r = std::nextafterf(r, -INFINITY);
    return r;
}

float slots_to_norm(const uint32_t* s) {
    uint32_t m = 0;
    for (int i = 0; i < kSlots; ++i) m = s[i] > m ? s[i] : m;
    float f;
    std::memcpy(&f, &m, 4);
    return host_sqrt_rd(f);
}

}  // namespace

struct sobfu_hip_solver {
    int X = 0, Y = 0, Z = 0;
    size_t N = 0;
    sobfu_hip_solver_params p{};
    float taps[7]{};
    // device workspace
    float* nabla_U     = nullptr;  // 16 B/voxel
    float* updates     = nullptr;  // 16 B/voxel, allocated on first need (verbosity > 0 or keep_updates)
    // compact iteration state (12-byte psi, 4-byte tsdf-only phi_global / phi_n / phi_n o psi), allocated on first
    // quiet solve; nabla_U doubles as the 12-byte nabla_U buffer
    float* c_psi = nullptr;  // 12 B/voxel
    float* c_f   = nullptr;  //  4 B/voxel  (phi_n o psi).tsdf
    float* c_g   = nullptr;  //  4 B/voxel  phi_global.tsdf
    float* c_n   = nullptr;  //  4 B/voxel  phi_n.tsdf
    bool compact = true;
    uint32_t* slots    = nullptr;  // (slots_iters + 1) x 256
    uint32_t* h_rows   = nullptr;  // pinned mirror of the slot rows for the non-blocking convergence poll
    int h_rows_iters   = 0;
    hipEvent_t ev_chk  = nullptr;
    void* red_scratch  = nullptr;  // 65536 x 8 B block partials
    int slots_iters    = 0;
    bool keep_updates  = false;
    sobfu_hip_log_fn log_fn = nullptr;
    void* log_user          = nullptr;
    bool log_set            = false;
    size_t bytes            = 0;
    // optional per-kernel timing (HIP events on the solver's stream around the launches of every prof_stride-th iteration)
    int prof_stride = 0;
    std::vector<hipEvent_t> events;
    double ms_a = 0, ms_b = 0;
    int prof_launches = 0, prof_pending = 0;
    // an open quiet solve (session_begin .. session_end)
    struct Session {
        bool active = false, compact = false;
        const float *pg = nullptr, *pn = nullptr;
        float *pnp = nullptr, *psi = nullptr;
        const float *it_pnp = nullptr, *it_pg = nullptr, *it_pn = nullptr;  // what the kernels iterate on
        float *it_psi = nullptr, *it_out = nullptr, *upd = nullptr;
        int cap = 0;       // iterations this session may run (slot rows cleared by begin)
        int launched = 0;  // iterations enqueued so far
        int checked = 0, fl_to = 0;  // convergence poll: rows examined / rows of the copy in flight
        bool in_flight = false;
        int done = 0;  // iterations known to have executed
        bool converged = false;
        float last_norm = 0.f;
        bool inline_log = false;  // the verbose loop prints its lines as it goes; session_end then only adds the closing line
    } q;

    void log(const std::string& line) const {
        if (log_set) {
            if (log_fn) log_fn(line.c_str(), log_user);
        } else {
            std::printf("%s\n", line.c_str());  // the reference writes to std::cout
        }
    }
};

namespace {

std::string fmt_g(float v) {  // std::cout default float formatting (%g, precision 6)
    char b[64];
    std::snprintf(b, sizeof b, "%g", (double) v);
    return b;
}

int ensure_slots(sobfu_hip_solver* s, int iters) {
    if (iters <= s->slots_iters) return 0;
    if (s->slots) {
        SOBFU_HIP_TRY(hipFree(s->slots));
        s->bytes -= (size_t) (s->slots_iters + 1) * kSlots * 4;
    }
    s->slots = nullptr;
    SOBFU_HIP_TRY(hipMalloc((void**) &s->slots, (size_t) (iters + 1) * kSlots * 4));
    s->slots_iters = iters;
    s->bytes += (size_t) (iters + 1) * kSlots * 4;
    return 0;
}

// pinned host mirror + event for the convergence poll (rows are copied asynchronously; the host never drains the stream
// just to look at them)
int ensure_poll(sobfu_hip_solver* s, int iters) {
    if (!s->ev_chk) SOBFU_HIP_TRY(hipEventCreateWithFlags(&s->ev_chk, hipEventDisableTiming));
    if (iters <= s->h_rows_iters) return 0;
    if (s->h_rows) SOBFU_HIP_TRY(hipHostFree(s->h_rows));
    s->h_rows = nullptr;
    SOBFU_HIP_TRY(hipHostMalloc((void**) &s->h_rows, (size_t) iters * kSlots * 4, hipHostMallocDefault));
    s->h_rows_iters = iters;
    return 0;
}

int ensure_compact(sobfu_hip_solver* s) {
    if (s->c_psi) return 0;
    SOBFU_HIP_TRY(hipMalloc((void**) &s->c_psi, s->N * 12));
    SOBFU_HIP_TRY(hipMalloc((void**) &s->c_f, s->N * 4));
    SOBFU_HIP_TRY(hipMalloc((void**) &s->c_g, s->N * 4));
    SOBFU_HIP_TRY(hipMalloc((void**) &s->c_n, s->N * 4));
    s->bytes += s->N * 24;
    return 0;
}

int ensure_updates(sobfu_hip_solver* s) {
    if (s->updates) return 0;
    SOBFU_HIP_TRY(hipMalloc((void**) &s->updates, s->N * 16));
    s->bytes += s->N * 16;
    return 0;
}

int set_params(sobfu_hip_solver* s, const sobfu_hip_solver_params* p) {
    SOBFU_CHECK_ARGS(p && p->max_iter >= 0);
    // The kernels use 7 taps whatever s is (KERNEL_RADIUS 3, solver.cu:211-234: cudaMemcpyToSymbol copies the first
    // 7 floats of the s-tap table).  s < 7 would read past the reference's allocation -> refused.
    if (p->s < 7 || p->s > 16) return p->s < 7 ? SOBFU_E_UNSUPPORTED : SOBFU_E_FILTER;
    float h[16];
    SOBFU_TRY(sobfu_hip_sobolev_filter(p->s, p->lambda, h));
    for (int i = 0; i < 7; ++i) s->taps[i] = h[i];
    s->p = *p;
    return 0;
}

int ensure_events(sobfu_hip_solver* s, size_t n) {
    while (s->events.size() < n) {
        hipEvent_t e;
        SOBFU_HIP_TRY(hipEventCreate(&e));
        s->events.push_back(e);
    }
    return 0;
}

// ---- the quiet (verbosity 0) loop in three pieces: begin / enqueue / end ------------------------------------------------
// begin: enters the iteration format (compact: pack psi + tsdf channels + the warp of solver.cu:106; API format: the warp),
//        clears the max-norm rows.  enqueue: launches pass A + pass B per iteration (asynchronous; the device-side gate makes
//        every launch after the reference's `break` a no-op).  end: reads the rows, finds the iteration the reference stops at,
//        rebuilds the caller's buffers.  sobfu_hip_solver_iterate / estimate_psi are begin + enqueue(max_iter) + end;
//        the session entry points of the C ABI expose the pieces (a frame loop that interleaves other work, bench.py's
//        timed region of exactly K iterations).
int session_begin_impl(sobfu_hip_solver* s, const float* pg, const float* pn, float* pnp, float* psi, int cap, hipStream_t st) {
    sobfu_hip_solver::Session& q = s->q;
    const int X = s->X, Y = s->Y, Z = s->Z;
    q = sobfu_hip_solver::Session{};
    q.pg = pg; q.pn = pn; q.pnp = pnp; q.psi = psi;
    q.cap = cap;
    q.compact = s->compact && cap > 0;
    q.last_norm = NAN;
    if (!q.compact) SOBFU_TRY(sobfu_hip_apply(pn, pnp, psi, X, Y, Z, st));  // solver.cu:106
    if (cap <= 0) return 0;
    SOBFU_TRY(ensure_slots(s, cap));
    SOBFU_HIP_TRY(hipMemsetAsync(s->slots, 0, (size_t) (cap + 1) * kSlots * 4, st));
    if (s->keep_updates) {
        SOBFU_TRY(ensure_updates(s));
        q.upd = s->updates;
    }
    // compact mode: iterate on private 12-byte psi / nabla_U and tsdf-only TSDF copies; the API buffers are rebuilt by
    // session_end (psi.xyz written back, phi_n o psi = apply(phi_n, psi) once) -- same values as iterating in place
    q.it_pnp = pnp; q.it_pg = pg; q.it_pn = pn; q.it_psi = psi; q.it_out = pnp;
    if (q.compact) {
        SOBFU_TRY(ensure_compact(s));
        SOBFU_TRY(sobfu_hip::launch_compact_enter(psi, pg, pn, s->c_psi, s->c_g, s->c_n, s->c_f, X, Y, Z, st));  // incl. solver.cu:106
        q.it_pnp = s->c_f; q.it_pg = s->c_g; q.it_pn = s->c_n; q.it_psi = s->c_psi; q.it_out = s->c_f;
    }
    if (s->p.max_update_norm >= 0.f) SOBFU_TRY(ensure_poll(s, cap));
    if (s->prof_stride > 0) SOBFU_TRY(ensure_events(s, (size_t) 3 * (cap / s->prof_stride + 1)));
    return 0;
}
// the session is open only once every allocation and launch of begin has succeeded: a failed begin (out of memory at 512^3,
// say) leaves the handle usable instead of rejecting every later call with "session open"
int session_begin(sobfu_hip_solver* s, const float* pg, const float* pn, float* pnp, float* psi, int cap, hipStream_t st) {
    if (s->q.active) return SOBFU_E_BADARG;
    const int rc = session_begin_impl(s, pg, pn, pnp, psi, cap, st);
    s->q.active = rc == 0;
    return rc;
}

// examine the rows [q.checked, upto) that sit in the pinned mirror
void session_examine(sobfu_hip_solver* s, int upto, float* per_iter) {
    sobfu_hip_solver::Session& q = s->q;
    for (int k = q.checked; k < upto && !q.converged; ++k) {
        const float v = slots_to_norm(s->h_rows + (size_t) k * kSlots);
        if (per_iter) per_iter[k] = v;
        q.last_norm = v;
        q.done = k + 1;
        if (v <= s->p.max_update_norm) q.converged = true;  // solver.cu:183 -- later launches were device-side no-ops
    }
    q.checked = upto;
}

int session_enqueue(sobfu_hip_solver* s, int n, bool poll, float* per_iter, hipStream_t st) {
    sobfu_hip_solver::Session& q = s->q;
    if (!q.active || n < 0 || q.launched + n > q.cap) return SOBFU_E_BADARG;
    const int X = s->X, Y = s->Y, Z = s->Z;
    const sobfu_hip_solver_params& p = s->p;
    const bool can_converge = p.max_update_norm >= 0.f;  // ||u|| >= 0 > negative threshold: never fires
    const int last = q.launched + n;
    for (int it = q.launched + 1; it <= last && !q.converged; ++it) {
        const uint32_t* prev = (it > 1) ? s->slots + (size_t) (it - 1) * kSlots : nullptr;
        uint32_t* cur        = s->slots + (size_t) it * kSlots;
        // timing events (sobfu_hip_solver_set_profiling): an event between two kernels costs a few us of drained pipeline, so
        // a profiled run is not the run whose wall time is quoted
        const bool ev = s->prof_stride > 0 && (it % s->prof_stride == 0) && (size_t) 3 * (s->prof_pending + 1) <= s->events.size();
        const int e0  = 3 * s->prof_pending;
        if (ev) SOBFU_HIP_TRY(hipEventRecord(s->events[e0], st));
        SOBFU_TRY(sobfu_hip::launch_pass_a(q.it_pnp, q.it_pg, q.it_psi, s->nabla_U, p.w_reg, X, Y, Z, prev, p.max_update_norm, 0, st, q.compact));
        if (ev) SOBFU_HIP_TRY(hipEventRecord(s->events[e0 + 1], st));
        SOBFU_TRY(sobfu_hip::launch_pass_b(s->nabla_U, q.it_psi, q.it_pn, q.it_out, q.upd, cur, s->taps, p.alpha, X, Y, Z, prev,
                                           p.max_update_norm, 0, st, 0, 0, 0, q.compact));
        if (ev) {
            SOBFU_HIP_TRY(hipEventRecord(s->events[e0 + 2], st));
            s->prof_pending += 1;
        }
        q.launched = it;
        if (can_converge && poll) {
            // the host looks at the max-norm rows WITHOUT draining the stream: every kCheckEvery iterations the finished
            // rows are copied to pinned memory behind the kernels, and the copy's event is polled; the host may run
            // at most kMaxRunAhead iterations past the last row it has seen (launches after the break are no-ops,
            // but each still costs a few us)
            if (!q.in_flight && (it % kCheckEvery == 0 || it == last)) {
                SOBFU_HIP_TRY(hipMemcpyAsync(s->h_rows + (size_t) q.checked * kSlots, s->slots + (size_t) (q.checked + 1) * kSlots,
                                             (size_t) (it - q.checked) * kSlots * 4, hipMemcpyDeviceToHost, st));
                SOBFU_HIP_TRY(hipEventRecord(s->ev_chk, st));
                q.in_flight = true;
                q.fl_to     = it;
            }
            if (q.in_flight) {
                const bool must_wait = it - q.checked >= kMaxRunAhead;
                hipError_t e = must_wait ? hipEventSynchronize(s->ev_chk) : hipEventQuery(s->ev_chk);
                if (e == hipSuccess) {
                    session_examine(s, q.fl_to, per_iter);
                    q.in_flight = false;
                } else if (e != hipErrorNotReady) {
                    return (int) e;
                }
            }
        }
    }
    return 0;
}

// examine every max-norm row launched so far (synchronises the stream).  With a threshold that can fire, rows are examined in
// order until the reference's break; otherwise all rows are read back at once.
int session_drain(sobfu_hip_solver* s, float* per_iter, hipStream_t st) {
    sobfu_hip_solver::Session& q = s->q;
    if (q.launched == 0) return 0;
    if (s->p.max_update_norm >= 0.f) {
        if (q.in_flight) {
            SOBFU_HIP_TRY(hipEventSynchronize(s->ev_chk));
            session_examine(s, q.fl_to, per_iter);
            q.in_flight = false;
        }
        if (!q.converged && q.checked < q.launched) {  // rows launched after the last copy was issued
            SOBFU_HIP_TRY(hipMemcpyAsync(s->h_rows + (size_t) q.checked * kSlots, s->slots + (size_t) (q.checked + 1) * kSlots,
                                         (size_t) (q.launched - q.checked) * kSlots * 4, hipMemcpyDeviceToHost, st));
            SOBFU_HIP_TRY(hipStreamSynchronize(st));
            session_examine(s, q.launched, per_iter);
        }
    } else {
        std::vector<uint32_t> hs((size_t) q.launched * kSlots);
        SOBFU_HIP_TRY(hipMemcpyAsync(hs.data(), s->slots + kSlots, hs.size() * 4, hipMemcpyDeviceToHost, st));
        SOBFU_HIP_TRY(hipStreamSynchronize(st));
        for (int k = 0; k < q.launched; ++k) {
            const float v = slots_to_norm(hs.data() + (size_t) k * kSlots);
            if (per_iter) per_iter[k] = v;
            q.last_norm = v;
        }
        q.done = q.launched;
    }
    return 0;
}

int session_end(sobfu_hip_solver* s, sobfu_hip_solver_report* rep, float* per_iter, hipStream_t st) {
    sobfu_hip_solver::Session& q = s->q;
    if (!q.active) return SOBFU_E_BADARG;
    sobfu_hip_solver_report r{};
    r.last_max_update_norm = NAN;
    r.last_max_update_index = NAN;
    r.last_e_data = r.last_e_reg = NAN;
    SOBFU_TRY(session_drain(s, per_iter, st));
    // `done` iterations actually executed (later launches were no-ops): psi.xyz back + phi_n o psi = apply(phi_n, psi), the
    // state solver.cu:168 leaves behind, in one pass
    if (q.compact) SOBFU_TRY(sobfu_hip::launch_compact_leave(s->c_psi, q.pn, q.psi, q.pnp, s->X, s->Y, s->Z, st));
    // the lines the reference prints at verbosity 0 (solver.cu:115-117,184,189), emitted after the fact
    for (int it = 1; it <= q.done && !q.inline_log; ++it)
        if (it == 1 || it % 50 == 0) s->log("iter. no. " + std::to_string(it));
    if (q.converged) s->log("SOLVER CONVERGED AFTER " + std::to_string(q.done) + " ITERATIONS");
    else if (q.cap > 0 && q.done == q.cap) s->log("SOLVER REACHED MAX. NO. OF ITERATIONS WITHOUT CONVERGING");
    r.iterations = q.done;
    r.converged  = q.converged ? 1 : 0;
    r.last_max_update_norm = q.last_norm;
    q.active = false;
    SOBFU_HIP_TRY(hipStreamSynchronize(st));
    if (rep) *rep = r;
    return 0;
}

// verbose (verbosity > 0).  The reference evaluates the two energies and prints the arg-max of the update only on REPORTING
// iterations -- every one at verbosity 2; 1, 50 k and max_iter at verbosity 1 (solver.cu:132-142,173-181).  Everything between two
// reporting iterations runs exactly like the quiet loop (iteration format, no host sync, device-side convergence gate); a
// reporting iteration stays in the iteration format too: energies straight from the tsdf-only / 12-byte arrays, the pass B that
// also stores `updates`, and the reference's arg-max reduction over them.  Lines, arrays and the iteration the solver stops at
// are the reference's.
int run_verbose(sobfu_hip_solver* s, const float* pg, const float* pn, float* pnp, float* psi, int max_iter,
                sobfu_hip_solver_report* rep, float* per_iter, hipStream_t st) {
    const int X = s->X, Y = s->Y, Z = s->Z;
    const sobfu_hip_solver_params& p = s->p;
    auto reports = [&](int it) { return p.verbosity == 2 || it == 1 || it % 50 == 0 || it == max_iter; };
    SOBFU_TRY(ensure_updates(s));
    SOBFU_TRY(session_begin(s, pg, pn, pnp, psi, max_iter, st));
    sobfu_hip_solver::Session& q = s->q;
    q.inline_log = true;
    float e_data = NAN, e_reg = NAN, arg = NAN;
    auto fail = [&](int rc) {
        q.active = false;
        return rc;
    };
#define SOBFU_VTRY(expr)                    \
    do {                                    \
        const int rc_ = (int) (expr);       \
        if (rc_ != 0) return fail(rc_);     \
    } while (0)
    for (int it = 1; it <= max_iter && !q.converged;) {
        if (!reports(it)) {  // a run of quiet iterations it .. to
            int to = it;
            while (to + 1 <= max_iter && !reports(to + 1)) ++to;
            SOBFU_VTRY(session_enqueue(s, to - it + 1, true, per_iter, st));
            if (p.max_update_norm >= 0.f) SOBFU_VTRY(session_drain(s, per_iter, st));  // did the break fire inside the run?
            it = to + 1;
            continue;
        }
        if (it == 1 || it % 50 == 0) s->log("iter. no. " + std::to_string(it));
        if (q.compact) {  // solver.cu:132-142 (J of the displacement is rebuilt in registers, not stored)
            SOBFU_VTRY(sobfu_hip::data_energy_tsdf(q.it_pg, q.it_pnp, (int) s->N, s->red_scratch, &e_data, st));
            SOBFU_VTRY(sobfu_hip::reg_energy_from_psi3(q.it_psi, X, Y, Z, s->red_scratch, &e_reg, st));
        } else {
            SOBFU_VTRY(sobfu_hip_data_energy(pg, pnp, (int) s->N, s->red_scratch, &e_data, st));
            SOBFU_VTRY(sobfu_hip_reg_energy_sobolev_from_psi(psi, X, Y, Z, s->red_scratch, &e_reg, st));
        }
        const float e = e_data + p.w_reg * e_reg;
        s->log("data energy + w_reg * reg energy = " + fmt_g(e_data) + " + " + fmt_g(p.w_reg) + " * " + fmt_g(e_reg) + " = " + fmt_g(e));
        uint32_t* cur = s->slots + (size_t) it * kSlots;  // the row the next quiet iteration's gate reads
        SOBFU_VTRY(sobfu_hip::launch_pass_a(q.it_pnp, q.it_pg, q.it_psi, s->nabla_U, p.w_reg, X, Y, Z, nullptr, 0.f, 0, st, q.compact));
        SOBFU_VTRY(sobfu_hip::launch_pass_b(s->nabla_U, q.it_psi, q.it_pn, q.it_out, s->updates, cur, s->taps, p.alpha, X, Y, Z, nullptr, 0.f, 0, st, 0,
                                            0, 0, q.compact));
        float mx[2];
        SOBFU_VTRY(sobfu_hip_max_update_norm(s->updates, (int) s->N, s->red_scratch, mx, st));  // solver.cu:172 (synchronises)
        if (per_iter) per_iter[it - 1] = mx[0];
        arg = mx[1];
        q.launched = q.checked = q.done = it;
        q.last_norm = mx[0];
        {  // solver.cu:175-180 (index arithmetic reproduced as written)
            int ix = (int) (mx[1] / (float) (X * Y));
            int iy = (int) ((mx[1] - (float) (ix * X * Y)) / (float) X);
            int iz = (int) (mx[1] - (float) (X * (iy + Y * ix)));
            s->log("max. update norm " + fmt_g(mx[0]) + " at voxel (" + std::to_string(iz) + ", " + std::to_string(iy) + ", " + std::to_string(ix) + ")");
        }
        if (mx[0] <= p.max_update_norm) q.converged = true;  // solver.cu:183
        ++it;
    }
#undef SOBFU_VTRY
    const bool last_reported = q.done > 0 && reports(q.done);
    sobfu_hip_solver_report r{};
    SOBFU_TRY(session_end(s, &r, per_iter, st));
    r.last_e_data = e_data;  // of the last reporting iteration
    r.last_e_reg  = e_reg;
    r.last_max_update_index = last_reported ? arg : NAN;  // the arg-max exists only where the reference would have printed it
    if (rep) *rep = r;
    return 0;
}

// The gradient-descent loop.  Returns the number of iterations executed in rep->iterations.
int run_loop(sobfu_hip_solver* s, const float* pg, const float* pn, float* pnp, float* psi, int max_iter,
             sobfu_hip_solver_report* rep, float* per_iter, hipStream_t st) {
    if (s->q.active) return SOBFU_E_BADARG;  // a session is open on this handle
    if (s->p.verbosity > 0 && max_iter > 0) return run_verbose(s, pg, pn, pnp, psi, max_iter, rep, per_iter, st);
    SOBFU_TRY(session_begin(s, pg, pn, pnp, psi, max_iter, st));
    int rc = session_enqueue(s, max_iter, true, per_iter, st);
    if (rc != 0) {
        s->q.active = false;
        return rc;
    }
    return session_end(s, rep, per_iter, st);
}

}  // namespace

extern "C" {

int sobfu_hip_abi_version(void) { return SOBFU_HIP_ABI_VERSION; }

const char* sobfu_hip_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case SOBFU_E_BADARG: return "sobfu_hip: bad argument";
        case SOBFU_E_FILTER: return "sobfu_hip: (s, lambda) not in the Sobolev filter table";
        case SOBFU_E_UNSUPPORTED: return "sobfu_hip: unsupported configuration";
        case SOBFU_E_RCCL: return "sobfu_hip: RCCL not loaded or an RCCL call failed (see stderr)";
        case SOBFU_E_TIMEOUT: return "sobfu_hip: a peer rank did not answer within the deadline (see stderr); the tiled handle is dead";
        default: return code > 0 ? hipGetErrorString((hipError_t) code) : "sobfu_hip: unknown error";
    }
}

// decompose_sobolev_filter -- src/sobfu/solver.cpp:160-262
int sobfu_hip_sobolev_filter(int s, float lambda, float* h) {
    SOBFU_CHECK_ARGS(h);
    bool ok = false;
    if (s == 3 && lambda == 0.1f) { h[0] = 0.06537f; h[1] = 0.99572f; h[2] = h[0]; ok = true; }
    if (s == 7) {
        if (lambda == 0.05f) { h[0] = 0.00006f; h[1] = 0.00015f; h[2] = 0.03917f; h[3] = 0.99846f; ok = true; }
        if (lambda == 0.1f) { h[0] = 0.00030f; h[1] = 0.00441f; h[2] = 0.06571f; h[3] = 0.99565f; ok = true; }
        if (lambda == 0.2f) { h[0] = 0.00120f; h[1] = 0.01094f; h[2] = 0.10204f; h[3] = 0.98941f; ok = true; }
        if (lambda == 0.4f) { h[0] = 0.00169f; h[1] = 0.01312f; h[2] = 0.10927f; h[3] = 0.98781f; ok = true; }
        if (ok) { h[4] = h[2]; h[5] = h[1]; h[6] = h[0]; }
    }
    if (s == 9) {
        if (lambda == 0.05f) { h[0] = 0.000003f; h[1] = 0.00006f; h[2] = 0.00155f; h[3] = 0.03917f; h[4] = 0.99846f; ok = true; }
        if (lambda == 0.1f) { h[0] = 0.00002f; h[1] = 0.00030f; h[2] = 0.00441f; h[3] = 0.06571f; h[4] = 0.99565f; ok = true; }
        if (ok) { h[5] = h[3]; h[6] = h[2]; h[7] = h[1]; h[8] = h[0]; }
    }
    if (s == 11 && lambda == 0.1f) {
        h[0] = 0.0000015f; h[1] = 0.00002f; h[2] = 0.00030f; h[3] = 0.00441f; h[4] = 0.06571f; h[5] = 0.99565f;
        h[6] = h[4]; h[7] = h[3]; h[8] = h[2]; h[9] = h[1]; h[10] = h[0];
        ok = true;
    }
    if (!ok) return SOBFU_E_FILTER;  // the reference would run on an uninitialised filter
    volatile float sum = 0.f;
    for (int i = 0; i < s; ++i) sum = sum + h[i];
    for (int i = 0; i < s; ++i) h[i] = h[i] / sum;
    return 0;
}

int sobfu_hip_solver_create(sobfu_hip_solver** out, int X, int Y, int Z, const sobfu_hip_solver_params* params) {
    SOBFU_CHECK_ARGS(out && params && X > 1 && Y > 1 && Z > 1);
    if ((size_t) X * Y * Z > (size_t) 0x7fffffff) return SOBFU_E_UNSUPPORTED;  // int32 linear indices, like the reference
    auto* s = new sobfu_hip_solver();
    s->X = X; s->Y = Y; s->Z = Z;
    s->N = (size_t) X * Y * Z;
    if (const char* e = getenv("SOBFU_COMPACT")) s->compact = atoi(e) != 0;  // tuning override
    int rc = set_params(s, params);
    if (rc == 0) rc = (int) hipMalloc((void**) &s->nabla_U, s->N * 16);
    if (rc == 0) rc = (int) hipMalloc(&s->red_scratch, 65536 * 8);
    if (rc == 0) {
        s->bytes = s->N * 16 + 65536 * 8;
        rc = ensure_slots(s, params->max_iter > 0 ? params->max_iter : 1);
    }
    // the quiet path's compact state is part of the workspace from the start (like the reference's constructor, which
    // allocates everything up front): the first solve of a sequence pays no allocation
    if (rc == 0 && s->compact) rc = ensure_compact(s);
    if (rc == 0 && params->verbosity > 0) rc = ensure_updates(s);
    if (rc != 0) {
        sobfu_hip_solver_destroy(s);
        return rc;
    }
    *out = s;
    return 0;
}

int sobfu_hip_solver_destroy(sobfu_hip_solver* s) {
    if (!s) return 0;
    if (s->nabla_U) (void) hipFree(s->nabla_U);
    if (s->updates) (void) hipFree(s->updates);
    if (s->slots) (void) hipFree(s->slots);
    if (s->red_scratch) (void) hipFree(s->red_scratch);
    if (s->h_rows) (void) hipHostFree(s->h_rows);
    if (s->ev_chk) (void) hipEventDestroy(s->ev_chk);
    for (float* q : {s->c_psi, s->c_f, s->c_g, s->c_n})
        if (q) (void) hipFree(q);
    for (hipEvent_t e : s->events) (void) hipEventDestroy(e);
    delete s;
    return 0;
}

int sobfu_hip_solver_set_params(sobfu_hip_solver* s, const sobfu_hip_solver_params* params) {
    SOBFU_CHECK_ARGS(s);
    return set_params(s, params);
}

size_t sobfu_hip_solver_workspace_bytes(const sobfu_hip_solver* s) { return s ? s->bytes : 0; }

float* sobfu_hip_solver_updates(sobfu_hip_solver* s) {
    if (!s) return nullptr;
    if (ensure_updates(s) != 0) return nullptr;
    return s->updates;
}

int sobfu_hip_solver_keep_updates(sobfu_hip_solver* s, int keep) {
    SOBFU_CHECK_ARGS(s);
    s->keep_updates = keep != 0;
    return 0;
}

int sobfu_hip_solver_set_logger(sobfu_hip_solver* s, sobfu_hip_log_fn fn, void* user) {
    SOBFU_CHECK_ARGS(s);
    s->log_fn   = fn;
    s->log_user = user;
    s->log_set  = true;
    return 0;
}

int sobfu_hip_solver_set_compact(sobfu_hip_solver* s, int enable) {
    SOBFU_CHECK_ARGS(s);
    s->compact = enable != 0;
    return 0;
}

int sobfu_hip_solver_set_profiling(sobfu_hip_solver* s, int stride) {
    SOBFU_CHECK_ARGS(s && stride >= 0);
    s->prof_stride = stride;
    const int cap = s->q.active ? s->q.cap : (s->p.max_iter > 0 ? s->p.max_iter : 1);
    if (stride > 0) SOBFU_TRY(ensure_events(s, (size_t) 3 * (cap / stride + 1)));  // not inside a timed solve
    return 0;
}

int sobfu_hip_solver_get_profile(sobfu_hip_solver* s, float* ms_pass_a, float* ms_pass_b, int* launches, int reset) {
    SOBFU_CHECK_ARGS(s);
    for (int it = 1; it <= s->prof_pending; ++it) {  // events recorded since the last call (the caller has synchronised)
        float a = 0, b = 0;
        SOBFU_HIP_TRY(hipEventElapsedTime(&a, s->events[3 * (it - 1)], s->events[3 * (it - 1) + 1]));
        SOBFU_HIP_TRY(hipEventElapsedTime(&b, s->events[3 * (it - 1) + 1], s->events[3 * (it - 1) + 2]));
        s->ms_a += a;
        s->ms_b += b;
        s->prof_launches += 1;
    }
    s->prof_pending = 0;
    if (ms_pass_a) *ms_pass_a = (float) s->ms_a;
    if (ms_pass_b) *ms_pass_b = (float) s->ms_b;
    if (launches) *launches = s->prof_launches;
    if (reset) {
        s->ms_a = s->ms_b = 0;
        s->prof_launches = 0;
    }
    return 0;
}

int sobfu_hip_solver_iterate(sobfu_hip_solver* s, const float* d_phi_global, const float* d_phi_n, float* d_phi_n_psi,
                             float* d_psi, int n_iters, sobfu_hip_solver_report* report, float* per_iter_max_norm,
                             void* stream) {
    SOBFU_CHECK_ARGS(s && d_phi_global && d_phi_n && d_phi_n_psi && d_psi && n_iters >= 0);
    SOBFU_TRY(run_loop(s, d_phi_global, d_phi_n, d_phi_n_psi, d_psi, n_iters, report, per_iter_max_norm, (hipStream_t) stream));
    return (int) hipStreamSynchronize((hipStream_t) stream);
}

int sobfu_hip_solver_begin(sobfu_hip_solver* s, const float* d_phi_global, const float* d_phi_n, float* d_phi_n_psi, float* d_psi,
                           int max_iters, void* stream) {
    SOBFU_CHECK_ARGS(s && d_phi_global && d_phi_n && d_phi_n_psi && d_psi && max_iters >= 0);
    if (s->p.verbosity > 0) return SOBFU_E_UNSUPPORTED;  // the verbose loop synchronises every iteration: use estimate_psi / iterate
    return session_begin(s, d_phi_global, d_phi_n, d_phi_n_psi, d_psi, max_iters, (hipStream_t) stream);
}

int sobfu_hip_solver_step(sobfu_hip_solver* s, int n_iters, void* stream) {
    SOBFU_CHECK_ARGS(s && n_iters >= 0);
    return session_enqueue(s, n_iters, false, nullptr, (hipStream_t) stream);
}

int sobfu_hip_solver_end(sobfu_hip_solver* s, sobfu_hip_solver_report* report, float* per_iter_max_norm, void* stream) {
    SOBFU_CHECK_ARGS(s);
    return session_end(s, report, per_iter_max_norm, (hipStream_t) stream);
}

int sobfu_hip_solver_estimate_psi(sobfu_hip_solver* s, const float* d_phi_global, float* d_phi_global_psi_inv,
                                  const float* d_phi_n, float* d_phi_n_psi, float* d_psi, float* d_psi_inv,
                                  sobfu_hip_solver_report* report, float* per_iter_max_norm, void* stream) {
    SOBFU_CHECK_ARGS(s && d_phi_global && d_phi_global_psi_inv && d_phi_n && d_phi_n_psi && d_psi && d_psi_inv);
    hipStream_t st = (hipStream_t) stream;
    SOBFU_TRY(run_loop(s, d_phi_global, d_phi_n, d_phi_n_psi, d_psi, s->p.max_iter, report, per_iter_max_norm, st));
    // solver.cu:196-199 in one pass: psi^-1 <- identity, 48 sweeps, phi_global o psi^-1
    SOBFU_TRY(sobfu_hip_inverse_and_warp(d_psi, d_psi_inv, d_phi_global, d_phi_global_psi_inv, s->X, s->Y, s->Z, 48, st));
    return (int) hipStreamSynchronize(st);
}

}  // extern "C"
