// Test harness (CPU): the scalar steps of the registration loop, icp_amd/csrc/loop_steps.hh, as
// the product compiles them for the host, behind a C interface for tests/test_loop_steps_host.py.
// Built by __graft_entry__.build() (or by the test itself) into tests/_build/; not part of the
// product library.
#include <cstring>

#include "loop_steps.hh"

using namespace icp_dev;

extern "C"
{
    unsigned ls_state_size() { return (unsigned)sizeof(LoopState); }
    unsigned ls_iter_sums() { return (unsigned)kIterSums; }
    unsigned ls_apply_at() { return (unsigned)kApplyAt; }
    void ls_init(void* st, const double* sums4, const float* mu_m, float threshold,
                 unsigned max_iters, double n_total, int sharded)
    {
        std::memset(st, 0, sizeof(LoopState));
        loop_init_step(static_cast<LoopState*>(st), sums4, mu_m, threshold, max_iters, n_total,
                       sharded != 0);
    }
    void ls_solve(void* st, const double* sums12) { loop_solve_step(static_cast<LoopState*>(st), sums12); }
    void ls_finish(void* st, const double* sums4) { loop_finish_step(static_cast<LoopState*>(st), sums4); }
    void ls_advance(void* st, const double* sums16) { loop_advance_step(static_cast<LoopState*>(st), sums16); }
    int ls_nn_idle(const void* st) { return loop_nn_idle(static_cast<const LoopState*>(st)) ? 1 : 0; }
    void ls_get(const void* stv, float* T16, float* Tn12, float* mu_q3, float* mu_p3, unsigned* it,
                unsigned* started, unsigned* done, float* error, double* work3)
    {
        const LoopState* st = static_cast<const LoopState*>(stv);
        std::memcpy(T16, st->T, sizeof st->T);
        std::memcpy(Tn12, st->Tn, sizeof st->Tn);
        std::memcpy(mu_q3, st->mu_q, sizeof st->mu_q);
        std::memcpy(mu_p3, st->mu_p, sizeof st->mu_p);
        *it = st->it;
        *started = st->started;
        *done = st->done;
        *error = st->error;
        std::memcpy(work3, st->work, sizeof st->work);
    }
}
