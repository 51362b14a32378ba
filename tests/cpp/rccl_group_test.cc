// CPU test of the grouped weight broadcast's control flow (april_asr_amd/csrc/rccl_group.h) against a recording stub of the
// RCCL / HIP calls whose individual calls can be made to fail: the close-group-THEN-abort order of a failure inside the open group
// (APRIL_FAULT_RCCL=2 needs two real devices and has never run on hardware), a failing group end, a failing set-up, the good path.
// build + run: tests/test_rccl_group_cpu.py (g++ -std=c++17 -I april_asr_amd/csrc)
#include "rccl_group.h"
#include <cstdio>
#include <string>

struct Stub {
    typedef int Comm;                       // 0 = null
    std::string log;
    int fail_init = 0, fail_start = 0, fail_broadcast_at = -1, fail_end = 0, fail_sync_at = -1;
    int group_open = 0, aborted = 0;
    bool comm_init_all(Comm *c, int n, const int *) { log += "I"; if (fail_init) return false; for (int i = 0; i < n; ++i) c[i] = 100 + i; return true; }
    bool group_start() { log += "S"; if (fail_start) return false; group_open = 1; return true; }
    bool group_end() { log += group_open ? "E" : "e"; group_open = 0; return !fail_end && broadcast_failed == 0; }      // (a group that recorded an error reports it)
    int broadcast_failed = 0;
    bool broadcast(int i, Comm c) {
        log += "B" + std::to_string(i);
        if (!group_open || c != 100 + i) { log += "!"; return false; }
        if (i == fail_broadcast_at) { broadcast_failed = 1; return false; }
        return true;
    }
    void comm_abort(Comm c) { log += group_open ? "A!" : "A"; (void)c; ++aborted; }       // "A!" = aborted while the group was still open: the bug
    bool set_device(int) { return true; }
    bool stream_sync(int i) { log += "Y" + std::to_string(i); return i != fail_sync_at; }
    double now_ms() { return 1.0; }
};

static int g_bad = 0;
static void expect(const char *name, bool got, bool want, const std::string &log, const std::string &want_log, const std::vector<int> &comms, const std::vector<int> &want_comms)
{
    const bool ok = got == want && log == want_log && comms == want_comms;
    printf("%-34s %s  result %d log %s\n", name, ok ? "ok" : "FAILED", (int)got, log.c_str());
    if (!ok) { ++g_bad; printf("    wanted result %d log %s\n", (int)want, want_log.c_str()); }
}

int main()
{
    const std::vector<int> devs = {0, 1, 2};
    {   // good path: init, start, three broadcasts, end, three stream syncs; the communicators stay for the caller to destroy
        Stub s; std::vector<int> comms(3, 0); double t = 0;
        const bool r = aprilx::rccl_group_broadcast(s, devs, comms, &t);
        expect("good path", r, true, s.log, "ISB0B1B2EY0Y1Y2", comms, {100, 101, 102});
        if (t != 1.0) { ++g_bad; printf("    t_after_init not reported\n"); }
    }
    {   // the broadcast of peer 1 fails INSIDE the open group (APRIL_FAULT_RCCL=2): no further broadcast, the group is closed first,
        // then every communicator is aborted (none while the group is open) and nulled
        Stub s; s.fail_broadcast_at = 1; std::vector<int> comms(3, 0);
        const bool r = aprilx::rccl_group_broadcast(s, devs, comms, nullptr);
        expect("broadcast fails inside the group", r, false, s.log, "ISB0B1EAAA", comms, {0, 0, 0});
    }
    {   // the group end itself fails: abort everything (the group is closed by then)
        Stub s; s.fail_end = 1; std::vector<int> comms(3, 0);
        const bool r = aprilx::rccl_group_broadcast(s, devs, comms, nullptr);
        expect("group end fails", r, false, s.log, "ISB0B1B2EAAA", comms, {0, 0, 0});
    }
    {   // communicator set-up fails: nothing else is called, nothing to abort
        Stub s; s.fail_init = 1; std::vector<int> comms(3, 0);
        const bool r = aprilx::rccl_group_broadcast(s, devs, comms, nullptr);
        expect("communicator set-up fails", r, false, s.log, "I", comms, {0, 0, 0});
    }
    {   // the group cannot be opened: the communicators made so far are aborted (no group is open)
        Stub s; s.fail_start = 1; std::vector<int> comms(3, 0);
        const bool r = aprilx::rccl_group_broadcast(s, devs, comms, nullptr);
        expect("group start fails", r, false, s.log, "ISAAA", comms, {0, 0, 0});
    }
    {   // a stream fails after the collective was launched: reported, communicators left to the caller's destroy
        Stub s; s.fail_sync_at = 1; std::vector<int> comms(3, 0);
        const bool r = aprilx::rccl_group_broadcast(s, devs, comms, nullptr);
        expect("stream sync fails", r, false, s.log, "ISB0B1B2EY0Y1", comms, {100, 101, 102});
    }
    printf(g_bad ? "rccl_group_test: %d FAILED\n" : "rccl_group_test: all paths in order\n", g_bad);
    return g_bad ? 1 : 0;
}
