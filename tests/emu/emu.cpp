// tests/emu/emu.cpp -- fiber scheduler behind tests/emu/hip/hip_runtime.h (test infrastructure).
#include "hip/hip_runtime.h"
#include <sys/mman.h>

extern "C" void emu_switch(void** from_sp, void** to_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace emu {
static const size_t kStack = 256 * 1024;
static State g_state;
State& S() { return g_state; }

static void fiber_exit_bookkeeping() {
    State& s = g_state;
    Fiber& f = s.fibers[s.cur];
    f.done = true;
    s.blk_alive--;
    if (s.blk_alive > 0 && s.blk_arrived >= s.blk_alive) { s.blk_arrived = 0; s.blk_gen++; }
    State::Wave& w = s.waves[s.cur >> 6];
    w.alive--;
    if (w.alive > 0 && w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
}

extern "C" void emu_trampoline() {
    State& s = g_state;
    s.body(s.body_arg);
    fiber_exit_bookkeeping();
    Fiber& f = s.fibers[s.cur];
    emu_switch(&f.sp, &s.sched_sp);
    abort();
}

void yield_() {
    State& s = g_state;
    Fiber& f = s.fibers[s.cur];
    emu_switch(&f.sp, &s.sched_sp);
}

void run_block(void (*body)(void*), void* arg, dim3 grid, dim3 block, dim3 bidx, size_t shmem) {
    State& s = g_state;
    int n = (int)(block.x * block.y * block.z);
    if ((int)s.fibers.size() < n) {
        size_t old = s.fibers.size();
        s.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; i++) {
            s.fibers[i].stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE,
                                            MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (s.fibers[i].stack == MAP_FAILED) { perror("mmap"); abort(); }
        }
    }
    static std::vector<char> dyn;
    if (dyn.size() < shmem + 64) dyn.resize(shmem + 64);
    s.dyn_shared = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
    s.nthreads = n;
    s.body = body; s.body_arg = arg;
    s.gDim = grid; s.bDim = block; s.bIdx = bidx;
    s.blk_alive = n; s.blk_arrived = 0;
    s.waves.assign((n + 63) / 64, State::Wave());
    for (int i = 0; i < n; i++) {
        Fiber& f = s.fibers[i];
        f.done = false;
        f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        s.waves[i >> 6].alive++;
        // initial stack: 6 callee-saved slots + return address (trampoline); keep 16B alignment
        uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8);      // so that at trampoline entry rsp % 16 == 8 (as after a call)
        *--sp = (void*)&emu_trampoline;     // return address
        for (int k = 0; k < 6; k++) *--sp = nullptr;
        f.sp = sp;
    }
    int remaining = n;
    while (remaining > 0) {
        for (int i = 0; i < n; i++) {
            Fiber& f = s.fibers[i];
            if (f.done) continue;
            s.cur = i;
            emu_switch(&s.sched_sp, &f.sp);
            if (f.done) remaining--;
        }
    }
}
}  // namespace emu

// devutil.h reaches v_writelane_b32 through the LLVM intrinsic's name; here it is an ordinary function of that name
extern "C" int cjs_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
extern "C" int cjs_writelane(int val, int lane, int old) { return emu::lane() == (lane & 63) ? val : old; }
