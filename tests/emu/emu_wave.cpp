#include "emu_wave.h"
#include <cstdlib>

namespace {
constexpr int kLanes = 64;
constexpr int kMaxFibres = 512;
int g_nfibres = kLanes;
int g_arrived = 0;
long g_generation = 0;
constexpr size_t kStack = 1 << 20;
ucontext_t g_sched;
ucontext_t g_fibre[kMaxFibres];
bool g_done[kMaxFibres];
int g_cur = -1;
const std::function<void(int)> *g_body = nullptr;

void trampoline(int lane)
{
    (*g_body)(lane);
    g_done[lane] = true;
    swapcontext(&g_fibre[lane], &g_sched);
}
}  // namespace

int emu_lane() { return g_cur; }

void emu_yield()
{
    int lane = g_cur;
    swapcontext(&g_fibre[lane], &g_sched);
}

void emu_team_sync()
{
    const long gen = g_generation;
    if (++g_arrived == g_nfibres) {
        g_arrived = 0;
        g_generation++;
        return;
    }
    while (g_generation == gen) emu_yield();
}

void emu_run_team(int waves, const std::function<void(int, int)> &body)
{
    std::function<void(int)> flat = [&](int f) { body(f / kLanes, f % kLanes); };
    g_nfibres = waves * kLanes;
    g_arrived = 0;
    emu_run_wave(flat);
    g_nfibres = kLanes;
}

void emu_run_wave(const std::function<void(int)> &body)
{
    g_body = &body;
    const int kLanes = g_nfibres;
    std::vector<char *> stacks(kLanes);
    for (int l = 0; l < kLanes; l++) {
        stacks[l] = (char *)malloc(kStack);
        getcontext(&g_fibre[l]);
        g_fibre[l].uc_stack.ss_sp = stacks[l];
        g_fibre[l].uc_stack.ss_size = kStack;
        g_fibre[l].uc_link = &g_sched;
        g_done[l] = false;
        makecontext(&g_fibre[l], (void (*)())trampoline, 1, l);
    }
    bool any = true;
    while (any) {
        any = false;
        for (int l = 0; l < kLanes; l++) {
            if (g_done[l]) continue;
            g_cur = l;
            swapcontext(&g_sched, &g_fibre[l]);
            if (!g_done[l]) any = true;
        }
    }
    g_cur = -1;
    for (int l = 0; l < kLanes; l++) free(stacks[l]);
}
