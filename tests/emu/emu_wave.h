// emu_wave.h -- TEST INFRASTRUCTURE: run a wave-level kernel body on the CPU.
// 64 user-level fibres (ucontext) stand for the 64 lanes of one wavefront; WAVE_SYNC() in the
// kernel headers becomes emu_yield(), a round-robin switch, which is exactly a wave-wide barrier
// because all lanes execute the same sequence of sync points.  Nothing here is product code.
#pragma once
#include <ucontext.h>
#include <functional>
#include <vector>

void emu_yield();
int emu_lane();
// run body(lane) for lane = 0..63 as one wavefront
void emu_run_wave(const std::function<void(int)> &body);
// run body(wave, lane) for `waves` wavefronts of one work-group; emu_team_sync() is the work-group
// barrier (every fibre of the team must reach it), emu_yield() stays the per-wave barrier: the
// scheduler is round-robin over all fibres, so the lanes of a wave (which execute the same sequence
// of sync points) advance in lock step, and a fibre waiting at the team barrier simply keeps yielding.
void emu_run_team(int waves, const std::function<void(int, int)> &body);
void emu_team_sync();
