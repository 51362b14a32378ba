// The grouped weight broadcast of ONE process that drives several devices (APRIL_GPU_DEVICES=0,1,...; reference load site
// src/april_model.c:57-61), written against an API policy: the product instantiates it with RCCL + HIP (april_api.cc RealRccl),
// tests/cpp/rccl_group_test.cc with a recording stub whose calls can be made to fail -- the failure paths (a broadcast that reports an
// error inside the open group, a failing group end, a failing communicator set-up) can then be executed on a CPU, call order included.
// They need two real devices on a GPU box, which the test pool does not have (APRIL_FAULT_RCCL=2).
//
// Order on failure: some ranks have queued their part of the collective, the failing one has not.  A call that fails
// inside an open group records the error in the group: ending the group then discards what was queued and returns that error instead
// of launching a broadcast that would wait for the missing rank.  So: close the group FIRST (the queued tasks still point at live
// communicators), THEN abort every communicator; streams are left alone.  Aborted communicators are set to null: the caller destroys
// only what is left.
#pragma once
#include <cstddef>
#include <vector>

namespace aprilx {

// Api: typedef Comm; bool comm_init_all(Comm *comms, int n, const int *devs); bool group_start(); bool group_end();
//      bool broadcast(int peer, Comm comm); void comm_abort(Comm); bool set_device(int dev); bool stream_sync(int peer); double now_ms();
template <class Api>
bool rccl_group_broadcast(Api &api, const std::vector<int> &devs, std::vector<typename Api::Comm> &comms, double *t_after_init)
{
    const int n = (int)devs.size();
    if (!api.comm_init_all(comms.data(), n, devs.data())) return false;
    if (t_after_init) *t_after_init = api.now_ms();
    auto abort_all = [&]() { for (auto &c : comms) if (c) { api.comm_abort(c); c = typename Api::Comm(); } };
    if (!api.group_start()) { abort_all(); return false; }
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) ok = api.set_device(devs[(size_t)i]) && api.broadcast(i, comms[(size_t)i]);
    if (!ok) {
        (void)api.group_end();          // first: the group forgets what was queued (and reports the recorded error)
        abort_all();                    // then: no communicator survives a failed collective
        return false;
    }
    if (!api.group_end()) { abort_all(); return false; }
    for (int i = 0; i < n; ++i) if (!api.set_device(devs[(size_t)i]) || !api.stream_sync(i)) return false;
    return true;
}

}  // namespace aprilx
