#include "emu_wave.h"
#include <cstdlib>

namespace {
constexpr int kLanes = 64;
constexpr size_t kStack = 1 << 20;
ucontext_t g_sched;
ucontext_t g_fibre[kLanes];
bool g_done[kLanes];
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

void emu_run_wave(const std::function<void(int)> &body)
{
    g_body = &body;
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
