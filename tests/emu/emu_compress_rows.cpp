// emu_compress_rows.cpp — TEST INFRASTRUCTURE ONLY: runs the source of the row-mapped compress kernel
// (rust-lz-fear_amd/csrc/analysis/lz4_compress_rows.inc) on the CPU under the lock-step wavefront emulator of lzf_simt.h, so that the
// CPU suite (tests/test_emu_compress_rows.py) can compare the kernel's logic with the oracle without a GPU.  One fiber per lane,
// resumed round-robin at every cross-lane primitive; the waves of a launch run one after the other (they only share the job
// queue).  Nothing of this is in the product library; the product path fails without a HIP device.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../rust-lz-fear_amd/csrc/analysis/lz4_compress_rows.inc"

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
EmuWave* g_wave = nullptr;
const rows::Args* g_args = nullptr;

void lane_entry() {
    EmuWave* w = g_wave;
    const uint32_t my = w->cur;
    SimtEmu b{w, my};
    const uint64_t before = w->n_sync;
    (void)before;
    rows::compress_rows_wave(b, *g_args);
    w->finished[my] = true;
    // every lane passes the same number of lock-step points; the lanes behind this one are parked at the last of them
    for (;;) {
        const uint32_t nxt = my + 1u;
        if (nxt < 64u) { w->cur = nxt; lzf_emu_switch(&w->sp[my], w->sp[nxt]); }
        else lzf_emu_switch(&w->sp[my], w->sched_sp);
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

extern "C" int lzf_emu_compress_rows(const lzf_compress_job* jobs, lzf_job_result* results, uint32_t n_jobs, const uint32_t* perm,
                                     uint32_t n_waves, uint32_t rows_active, uint32_t alone, uint64_t* n_sync_out) {
    static uint8_t* stacks = nullptr;
    if (!stacks) stacks = (uint8_t*)malloc(kStack * 64);
    EmuWave* w = (EmuWave*)malloc(sizeof(EmuWave));
    if (!stacks || !w) return -1;
    uint32_t queue = 0;
    rows::Args a{jobs, results, n_jobs, perm, &queue, rows_active, alone};
    g_args = &a;
    uint64_t total = 0;
    int rc = 0;
    for (uint32_t wv = 0; wv < n_waves && rc == 0; ++wv) {
        memset(w, 0xA5, sizeof(EmuWave));                     // LDS starts as garbage, like on the device
        w->n_sync = 0; w->cur = 0;
        w->L = w->lds; w->next = w; w->wave_id = 0; w->n_waves = 1; w->bar_gen = 0; w->bar_count = &w->bar_gen;     // (a lone wave: its own LDS array, no barriers)
        for (uint32_t i = 0; i < 64; ++i) { w->sp[i] = make_stack(stacks + kStack * i); w->finished[i] = false; }
        g_wave = w;
        lzf_emu_switch(&w->sched_sp, w->sp[0]);
        for (uint32_t i = 0; i < 64; ++i) if (!w->finished[i]) { fprintf(stderr, "emu: lane %u did not finish (control flow not wave-uniform?)\n", i); rc = -2; }
        if (w->n_sync % 64u) { fprintf(stderr, "emu: %llu lock-step points are not a multiple of 64 lanes\n", (unsigned long long)w->n_sync); rc = -3; }
        total += w->n_sync / 64u;
    }
    if (n_sync_out) *n_sync_out = total;
    free(w);
    return rc;
}
