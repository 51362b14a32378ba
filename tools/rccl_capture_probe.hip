// Probe (round 6): may RCCL's communicator set-up and collectives run in one thread WITHOUT the process-wide
// hip_legacy_mutex (engine.h) while another thread captures graphs in relaxed mode?  Thread A captures + instantiates + launches small
// graphs on a non-blocking stream in a loop; thread B runs ncclCommInitRank (world 1) + ncclBroadcast + ncclCommDestroy in a loop.
// Control leg: thread B issues legacy-stream hipMemcpy calls instead (the case engine.h documents as poisoning captures).
// Prints the number of failed captures / failed RCCL calls per leg.
// build: hipcc --offload-arch=gfx950 -O2 tools/rccl_capture_probe.hip -L/opt/rocm/lib -lrccl -lpthread -o tools/rccl_capture_probe
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void bump(int *p) { atomicAdd(p, 1); }

static std::atomic<bool> g_stop{false};

static void capture_loop(int *dev_word, long *ok, long *bad)
{
    hipStream_t st;
    (void)hipSetDevice(0);
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { ++*bad; return; }
    while (!g_stop.load()) {
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        bool good = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess;
        for (int i = 0; i < 40 && good; ++i) hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, st, dev_word);
        const hipError_t e = hipStreamEndCapture(st, &graph);
        good = good && e == hipSuccess && graph != nullptr;
        if (good) good = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (good) good = hipGraphLaunch(exec, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (good) ++*ok; else ++*bad;
    }
    (void)hipStreamDestroy(st);
}

int main()
{
    (void)hipSetDevice(0);
    int *word = nullptr; (void)hipMalloc((void **)&word, 4); (void)hipMemset(word, 0, 4);
    float *buf = nullptr; (void)hipMalloc((void **)&buf, 64 << 20);
    for (int leg = 0; leg < 3; ++leg) {
        long ok = 0, bad = 0, rounds = 0, rbad = 0;
        g_stop = false;
        std::thread a(capture_loop, word, &ok, &bad);
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::steady_clock::now() - t0 < std::chrono::seconds(leg == 1 ? 12 : 4)) {
            if (leg == 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));          // baseline: nothing beside the captures
            else if (leg == 1) {                                                              // RCCL set-up + collective + tear-down
                ncclUniqueId id; ncclComm_t comm = nullptr; hipStream_t st = nullptr;
                bool good = ncclGetUniqueId(&id) == ncclSuccess && ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess;
                good = good && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
                good = good && ncclBroadcast(buf, buf, (64 << 20) / 4, ncclFloat, 0, comm, st) == ncclSuccess && hipStreamSynchronize(st) == hipSuccess;
                if (comm) (void)ncclCommDestroy(comm);
                if (st) (void)hipStreamDestroy(st);
                if (!good) ++rbad;
            } else {                                                                          // control: legacy-stream copies
                int h = 0;
                if (hipMemcpy(&h, word, 4, hipMemcpyDeviceToHost) != hipSuccess) { ++rbad; (void)hipGetLastError(); }
            }
            ++rounds;
        }
        g_stop = true; a.join();
        printf("leg %d (%s): captures ok %ld failed %ld | other thread: rounds %ld failed %ld\n", leg,
               leg == 0 ? "captures alone" : leg == 1 ? "RCCL init + broadcast + destroy beside the captures, no lock" : "legacy-stream hipMemcpy beside the captures, no lock", ok, bad, rounds, rbad);
        fflush(stdout);
    }
    return 0;
}
