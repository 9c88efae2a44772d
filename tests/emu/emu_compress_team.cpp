// emu_compress_team.cpp — TEST INFRASTRUCTURE ONLY: runs the source of the team compress kernel
// (rust-lz-fear_amd/csrc/lz4_compress_team.inc: searcher / emitter / feeder wavefronts of one workgroup per block) on the CPU under the
// lock-step wavefront emulator of lzf_simt.h, so that the CPU suite (tests/test_emu_compress_team.py) can compare the kernel's logic —
// the parse AND the flag protocol between its three waves — with the oracle without a GPU.  One fiber per lane; the lanes of a wave are
// resumed round-robin at every primitive, and behind the last lane of a wave the next wave of the workgroup runs up to ITS next
// primitive (one fixed, deterministic interleaving; the workgroups of a launch run one after the other).  Nothing of this is in the
// product library; the product path fails without a HIP device.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../rust-lz-fear_amd/csrc/lz4_compress_team.inc"

#if !defined(__x86_64__)
#error "the fiber switch below is x86-64 System V"
#endif
// callee-saved registers on the old stack, stack pointers swapped, callee-saved registers from the new stack
asm(R"(
.text
.globl lzf_emu_switch
.type lzf_emu_switch,@function
lzf_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size lzf_emu_switch,.-lzf_emu_switch
)");

namespace {
using namespace lzf;

constexpr size_t kStack = 256 * 1024;
constexpr uint32_t kWaves = 3;
const team::Args* g_args = nullptr;
uint32_t g_launch_index = 0;
uint32_t g_lanes_finished = 0;
void* g_sched_sp = nullptr;

void lane_entry() {
    EmuWave* w = emu_current();
    const uint32_t my = w->cur;
    SimtEmu b{w, my};
    // (alone: the round-5 kernel, compact jobs only; with the general kernel behind it the carry kernel takes the caller-owned tables —
    //  on the device both are launched, each leaving the other's jobs alone: here the one that owns the job runs)
    const lzf_compress_job& j = g_args->jobs[g_args->perm ? g_args->perm[g_launch_index] : g_launch_index];
    if (g_args->alone || compress_job_is_compact(j)) team::compress_team<SimtEmu, false>(b, *g_args, g_launch_index);
    else team::compress_team<SimtEmu, true>(b, *g_args, g_launch_index);
    w->finished[my] = true;
    ++g_lanes_finished;
    // a finished lane keeps handing the CPU on (the other waves are still running) until every lane of the workgroup is done
    for (;;) {
        if (g_lanes_finished == kWaves * 64u) lzf_emu_switch(&w->sp[my], g_sched_sp);
        if (my + 1u < 64u) { w->cur = my + 1u; lzf_emu_switch(&w->sp[my], w->sp[my + 1u]); }
        else { EmuWave* nx = w->next; nx->cur = 0; emu_current() = nx; lzf_emu_switch(&w->sp[my], nx->sp[0]); }
    }
}

void* make_stack(uint8_t* base) {          // a frame lzf_emu_switch can "return" into: six zero registers, then lane_entry
    uintptr_t top = (uintptr_t)(base + kStack);
    top &= ~(uintptr_t)15;
    top -= 16;                              // the slot of the return address at an address = 0 mod 16: after `ret` rsp = 8 mod 16, as after a call
    uint64_t* p = (uint64_t*)top;
    p[0] = (uint64_t)(uintptr_t)&lane_entry;
    p[1] = 0;
    for (int i = 1; i <= 6; ++i) p[-i] = 0;
    return (void*)(p - 6);
}
}  // namespace

extern "C" int lzf_emu_compress_team(const lzf_compress_job* jobs, lzf_job_result* results, uint32_t n_jobs, const uint32_t* perm,
                                     uint32_t alone, uint64_t* n_sync_out) {
    static uint8_t* stacks = nullptr;
    if (!stacks) stacks = (uint8_t*)malloc(kStack * 64 * kWaves);
    EmuWave* wv = (EmuWave*)malloc(sizeof(EmuWave) * kWaves);
    uint32_t* lds = (uint32_t*)malloc(sizeof(uint32_t) * team::kLdsWords + 64);
    if (!stacks || !wv || !lds) return -1;
    team::Args a{jobs, results, n_jobs, perm, alone};
    g_args = &a;
    uint64_t total = 0;
    int rc = 0;
    for (uint32_t j = 0; j < n_jobs && rc == 0; ++j) {
        memset(lds, 0xA5, sizeof(uint32_t) * team::kLdsWords);          // LDS starts as garbage, like on the device
        uint32_t bar_count = 0;
        for (uint32_t k = 0; k < kWaves; ++k) {
            EmuWave* w = &wv[k];
            w->n_sync = 0; w->cur = 0;
            w->L = lds; w->wave_id = k; w->next = &wv[(k + 1u) % kWaves]; w->bar_count = &bar_count; w->n_waves = kWaves; w->bar_gen = 0;
            for (uint32_t i = 0; i < 64; ++i) { w->sp[i] = make_stack(stacks + kStack * (64u * k + i)); w->finished[i] = false; }
        }
        g_launch_index = j;
        g_lanes_finished = 0;
        emu_current() = &wv[0];
        lzf_emu_switch(&g_sched_sp, wv[0].sp[0]);
        for (uint32_t k = 0; k < kWaves; ++k) {
            for (uint32_t i = 0; i < 64; ++i) if (!wv[k].finished[i]) { fprintf(stderr, "emu: wave %u lane %u did not finish\n", k, i); rc = -2; }
            if (wv[k].n_sync % 64u) { fprintf(stderr, "emu: wave %u: %llu lock-step points are not a multiple of 64 lanes (control flow not wave-uniform?)\n", k, (unsigned long long)wv[k].n_sync); rc = -3; }
            total += wv[k].n_sync / 64u;
        }
    }
    if (n_sync_out) *n_sync_out = total;
    free(wv); free(lds);
    return rc;
}
