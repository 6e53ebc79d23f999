// sdrx_group.hip -- one VFO tree on SEVERAL GPUs from ONE host process (included by sdrx.hip).
//
// The reference runs every VFO of every main on one thread (sdrj.cpp:288-294 -> vfo.cpp:253-264).
// Sub VFOs share nothing but their read-only input, so the tree shards: a group
// holds one sdrx_ctx per device, each with a static block of the sub VFOs of every main VFO (the unit
// that moves is a sub VFO with everything below it; a main VFO is replicated on every device that
// holds at least one of its subs, and on no other -- alone it would turn into an IQ-publishing leaf).
// The only exchange is the raw frame: it lands on the first device (host copy, or the caller's device
// buffer) and is fanned out to the others by hipMemcpyPeerAsync, one copy per peer on that peer's own
// stream -- over xGMI every peer is one direct link away from the source, so the N-1 copies run on
// N-1 different links at once (the flat broadcast SURVEY.md 5 asks for; a ring would be per-link bound
// and N-1 hops deep).  Frame f travels into buffer f & 1 while frame f-1 is still being processed.
// Payloads come back by each device's own D2H copy; the publish callback sees them in the REFERENCE's
// order (main order x sub order), whichever device computed them.
namespace {

struct GroupMember {
    sdrx_ctx *c = nullptr;
    int device = 0;
    std::vector<int> global_of; // local id -> global id
    unsigned char *d_frame[2] = {nullptr, nullptr}; // peers: the raw frame of parity p (cf32 or bytes)
};

} // namespace

struct sdrx_group {
    std::string err;
    std::vector<GroupMember> m;
    std::vector<sdrx_vfo_desc> descs;
    std::vector<std::pair<std::string, int>> options;
    std::vector<std::pair<int, int>> where; // global id -> (member, local id): the owner of a leaf, the FIRST replica of a VFO with children
    std::vector<int> publish_order;         // global ids of the leaves, reference order
    sdrx_publish_fn cb = nullptr;
    void *cb_user = nullptr;
    bool finalized = false;
    bool broken = false; // a launch or wait failed on one member after others had gone ahead: only destroy is left
    int root_frame = 0;
    int in_flight = 0;
    unsigned long long frame_no = 0;
    // staging of host-fed frames on the first device, per frame parity
    unsigned char *h_stage[2] = {nullptr, nullptr};
    unsigned char *d_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_ready[2] = {nullptr, nullptr}; // frame of parity p is complete on the first device
    hipStream_t stage_stream = nullptr;          // first device: host -> device copies of the raw frames
    bool peer_ok = true;
};

namespace {

int gfail(sdrx_group *g, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g)
        g->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define GHIP(g, expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return gfail((g), SDRX_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int member_fail(sdrx_group *g, int k, int rc)
{
    return gfail(g, rc, "device %d (member %d): %s", g->m[(size_t)k].device, k, sdrx_last_error(g->m[(size_t)k].c));
}

// frame (cf32 or bytes, `bytes` long) is complete on the first device at `src` once ev_ready[p] fires:
// fan it out and enqueue frame processing on every member.
int group_enqueue(sdrx_group *g, const void *src, size_t bytes, int raw_mode, bool egress, int correct_dc = 0)
{
    const int p = (int)(g->frame_no & 1ull);
    for (size_t k = 0; k < g->m.size(); ++k) {
        GroupMember &M = g->m[k];
        if (!M.c)
            continue;
        GHIP(g, hipSetDevice(M.device));
        const void *raw = src;
        GHIP(g, hipStreamWaitEvent(M.c->stream, g->ev_ready[p], 0));
        if (k > 0) {
            if (M.device != g->m[0].device)
                GHIP(g, hipMemcpyPeerAsync(M.d_frame[p], M.device, src, g->m[0].device, bytes, M.c->stream));
            else // two members on one device (tests on a 1-GPU box): the same transfer as a device copy
                GHIP(g, hipMemcpyAsync(M.d_frame[p], src, bytes, hipMemcpyDeviceToDevice, M.c->stream));
            raw = M.d_frame[p];
        }
        M.c->last_raw = -1;
        int rc;
        if (raw_mode == kRawU8 && correct_dc) {
            // sdrj.cpp:271-286 on every member: identical bytes and identical (zero-started) accumulators keep the
            // members' DC estimates identical, so no state ever has to travel between devices
            rc = enqueue_u8_device(M.c, raw, g->root_frame, 1, egress); // (leaves last_raw = tile layout: sdrx_get_raw serves it)
        } else {
            rc = enqueue_frame(M.c, raw, raw_mode, egress);
        }
        if (rc) {
            g->broken = true; // earlier members already run this frame
            return member_fail(g, (int)k, rc);
        }
    }
    g->frame_no++;
    if (egress)
        g->in_flight++;
    return SDRX_OK;
}

int group_check(sdrx_group *g, const char *what, const void *ptr, int n_complex, bool sync_call)
{
    if (!g)
        return SDRX_EINVAL;
    if (!ptr)
        return gfail(g, SDRX_EINVAL, "%s: null frame pointer", what);
    if (!g->finalized)
        return gfail(g, SDRX_ESTATE, "%s before sdrx_group_finalize", what);
    if (g->broken)
        return gfail(g, SDRX_ESTATE, "%s: an earlier call failed on one device after others had gone ahead; destroy the group", what);
    if (n_complex != g->root_frame)
        return gfail(g, SDRX_EINVAL, "frame of %d samples, VFOs were initialised for %d (vfo::init samplesPerBuffer)", n_complex, g->root_frame);
    if (sync_call && g->in_flight > 0)
        return gfail(g, SDRX_ESTATE, "%s: %d submitted frame(s) not yet delivered -- call sdrx_group_wait first", what, g->in_flight);
    if (!sync_call && g->in_flight >= SDRX_MAX_IN_FLIGHT)
        return gfail(g, SDRX_ESTATE, "%s: %d frames in flight -- call sdrx_group_wait before submitting another", what, g->in_flight);
    return SDRX_OK;
}

int group_stage(sdrx_group *g, const void *host, size_t bytes)
{
    const int p = (int)(g->frame_no & 1ull);
    GHIP(g, hipSetDevice(g->m[0].device));
    memcpy(g->h_stage[p], host, bytes);
    GHIP(g, hipMemcpyAsync(g->d_stage[p], g->h_stage[p], bytes, hipMemcpyHostToDevice, g->stage_stream));
    GHIP(g, hipEventRecord(g->ev_ready[p], g->stage_stream));
    return SDRX_OK;
}

// undo a partly done sdrx_group_finalize: a retry (or the destroy) starts from nothing
void group_drop_members(sdrx_group *g)
{
    for (GroupMember &M : g->m) {
        (void)hipSetDevice(M.device);
        if (M.c)
            sdrx_destroy(M.c);
        M.c = nullptr;
        M.global_of.clear();
        for (int p = 0; p < 2; ++p) {
            if (M.d_frame[p])
                (void)hipFree(M.d_frame[p]);
            M.d_frame[p] = nullptr;
        }
    }
    if (!g->m.empty())
        (void)hipSetDevice(g->m[0].device);
    for (int p = 0; p < 2; ++p) {
        if (g->h_stage[p])
            (void)hipHostFree(g->h_stage[p]);
        if (g->d_stage[p])
            (void)hipFree(g->d_stage[p]);
        if (g->ev_ready[p])
            (void)hipEventDestroy(g->ev_ready[p]);
        g->h_stage[p] = g->d_stage[p] = nullptr;
        g->ev_ready[p] = nullptr;
    }
    if (g->stage_stream)
        (void)hipStreamDestroy(g->stage_stream);
    g->stage_stream = nullptr;
    g->where.clear();
    g->publish_order.clear();
}

void group_publish(sdrx_group *g)
{
    if (!g->cb)
        return;
    for (int gid : g->publish_order) {
        const auto w = g->where[(size_t)gid];
        if (w.first < 0)
            continue;
        sdrx_ctx *c = g->m[(size_t)w.first].c;
        const Node &n = c->nodes[(size_t)w.second];
        if (n.pay_len == 0 || c->host_slot < 0)
            continue;
        if (!n.d.demod_usb && n.d.topic[0] == 0) // vfo::transmitData: an IQ leaf publishes only with a topic
            continue;
        char topic[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < 5 && n.d.topic[k]; ++k)
            topic[k] = n.d.topic[k];
        g->cb(g->cb_user, topic, n.rate, c->h_pay[c->host_slot] + n.pay_off, n.pay_len);
    }
}

} // namespace

extern "C" {

int sdrx_group_create(sdrx_group **out, const int *devices, int n_devices)
{
    if (!out || !devices || n_devices <= 0)
        return gfail(nullptr, SDRX_EINVAL, "sdrx_group_create: bad arguments");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return gfail(nullptr, SDRX_EHIP, "sdrx_group_create: no HIP device (%s); libsdrx has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= ndev)
            return gfail(nullptr, SDRX_EINVAL, "sdrx_group_create: device %d out of range (0..%d)", devices[k], ndev - 1);
    sdrx_group *g = new sdrx_group();
    g->m.resize((size_t)n_devices);
    for (int k = 0; k < n_devices; ++k)
        g->m[(size_t)k].device = devices[k];
    // peer access first device <-> every other one (hipMemcpyPeerAsync stages through the host without it)
    for (int k = 1; k < n_devices; ++k) {
        if (devices[k] == devices[0])
            continue;
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, devices[k], devices[0]);
        hipError_t pe = hipSuccess;
        if (can) {
            (void)hipSetDevice(devices[k]);
            pe = hipDeviceEnablePeerAccess(devices[0], 0);
            (void)hipGetLastError();
        }
        if (!can || (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)) {
            // not an error -- hipMemcpyPeerAsync still works, staged through host memory by the runtime -- but the
            // N-1 copies then no longer run on N-1 xGMI links at once: say so (sdrx_group_peer_access,
            // sdrx_group_last_error right after the create)
            g->peer_ok = false;
            char note[160];
            snprintf(note, sizeof note, "warning: device %d has no peer access to device %d%s%s: the raw-frame fan-out to it is host-staged",
                     devices[k], devices[0], can ? ": " : "", can ? hipGetErrorString(pe) : "");
            g->err = note;
        }
    }
    (void)hipSetDevice(devices[0]);
    *out = g;
    return SDRX_OK;
}

int sdrx_group_destroy(sdrx_group *g)
{
    if (!g)
        return SDRX_EINVAL;
    group_drop_members(g);
    delete g;
    return SDRX_OK;
}

const char *sdrx_group_last_error(const sdrx_group *g) { return g ? g->err.c_str() : g_create_error.c_str(); }

int sdrx_group_size(const sdrx_group *g) { return g ? (int)g->m.size() : SDRX_EINVAL; }

int sdrx_group_add_vfo(sdrx_group *g, const sdrx_vfo_desc *d, int *id_out)
{
    if (!g || !d)
        return SDRX_EINVAL;
    if (g->finalized)
        return gfail(g, SDRX_ESTATE, "sdrx_group_add_vfo after sdrx_group_finalize");
    const int id = (int)g->descs.size();
    if (d->parent_id >= id || d->parent_id < -1)
        return gfail(g, SDRX_EINVAL, "vfo %d: parent_id %d must name an earlier vfo or be -1", id, d->parent_id);
    g->descs.push_back(*d);
    if (id_out)
        *id_out = id;
    return SDRX_OK;
}

int sdrx_group_set_option(sdrx_group *g, const char *name, int value)
{
    if (!g || !name)
        return SDRX_EINVAL;
    if (g->finalized)
        return gfail(g, SDRX_ESTATE, "sdrx_group_set_option after sdrx_group_finalize");
    g->options.emplace_back(name, value);
    return SDRX_OK;
}

int sdrx_group_set_publish_callback(sdrx_group *g, sdrx_publish_fn fn, void *user)
{
    if (!g)
        return SDRX_EINVAL;
    g->cb = fn;
    g->cb_user = user;
    return SDRX_OK;
}

// The partition (the same rule as sdrreceiver_amd.topology.shard): for every parent-less VFO with
// children, member k of W gets children [K k / W, K (k+1) / W) with their whole subtrees and -- only if
// that block is not empty -- a replica of the parent; parent-less leaves are block-partitioned among
// themselves.
static int group_finalize_impl(sdrx_group *g);

int sdrx_group_finalize(sdrx_group *g)
{
    if (!g)
        return SDRX_EINVAL;
    if (g->finalized)
        return gfail(g, SDRX_ESTATE, "sdrx_group_finalize called twice");
    const int rc = group_finalize_impl(g);
    if (rc != SDRX_OK) { // nothing of a half-built group stays behind (the error text does)
        const std::string why = g->err;
        group_drop_members(g);
        g->err = why;
    }
    return rc;
}

static int group_finalize_impl(sdrx_group *g)
{
    const int N = (int)g->descs.size(), W = (int)g->m.size();
    if (N == 0)
        return gfail(g, SDRX_ESTATE, "sdrx_group_finalize: no VFOs");
    std::vector<std::vector<int>> children((size_t)N);
    std::vector<int> roots, flat;
    for (int i = 0; i < N; ++i) {
        if (g->descs[(size_t)i].parent_id >= 0)
            children[(size_t)g->descs[(size_t)i].parent_id].push_back(i);
        else
            roots.push_back(i);
    }
    // every parent-less VFO consumes the same raw frame (sdrj.cpp:288-294): one samples_per_buffer for all of
    // them, whichever member they land on -- the staging and peer buffers below are sized for it
    g->root_frame = g->descs[(size_t)roots[0]].samples_per_buffer;
    for (int r : roots) {
        if (g->descs[(size_t)r].samples_per_buffer != g->root_frame)
            return gfail(g, SDRX_EINVAL, "vfo %d: all parent-less VFOs must share samples_per_buffer (%d != %d)", r,
                         g->descs[(size_t)r].samples_per_buffer, g->root_frame);
        if (children[(size_t)r].empty())
            flat.push_back(r);
    }
    if (g->root_frame <= 0)
        return gfail(g, SDRX_EINVAL, "vfo %d: samples_per_buffer must be positive", roots[0]);
    g->where.assign((size_t)N, std::make_pair(-1, -1));
    for (int k = 0; k < W; ++k) {
        std::vector<char> keep((size_t)N, 0);
        std::vector<int> stack;
        for (int r : roots) {
            const std::vector<int> &ch = children[(size_t)r];
            if (ch.empty())
                continue;
            const size_t lo = ch.size() * (size_t)k / (size_t)W, hi = ch.size() * (size_t)(k + 1) / (size_t)W;
            if (hi <= lo)
                continue;
            keep[(size_t)r] = 1;
            for (size_t q = lo; q < hi; ++q)
                stack.push_back(ch[q]);
        }
        while (!stack.empty()) {
            const int j = stack.back();
            stack.pop_back();
            keep[(size_t)j] = 1;
            for (int ch : children[(size_t)j])
                stack.push_back(ch);
        }
        const size_t lo = flat.size() * (size_t)k / (size_t)W, hi = flat.size() * (size_t)(k + 1) / (size_t)W;
        for (size_t q = lo; q < hi; ++q)
            keep[(size_t)flat[q]] = 1;
        GroupMember &M = g->m[(size_t)k];
        std::vector<int> local((size_t)N, -1);
        int count = 0;
        for (int i = 0; i < N; ++i)
            count += keep[(size_t)i];
        if (count == 0)
            continue; // more devices than sub VFOs: this one holds nothing
        GHIP(g, hipSetDevice(M.device));
        int rc = sdrx_create(&M.c, M.device);
        if (rc)
            return gfail(g, rc, "device %d: %s", M.device, sdrx_last_error(nullptr));
        for (auto &o : g->options)
            if ((rc = sdrx_set_option(M.c, o.first.c_str(), o.second)) != SDRX_OK)
                return member_fail(g, k, rc);
        for (int i = 0; i < N; ++i) {
            if (!keep[(size_t)i])
                continue;
            sdrx_vfo_desc d = g->descs[(size_t)i];
            d.parent_id = d.parent_id >= 0 ? local[(size_t)d.parent_id] : -1;
            int lid = -1;
            if ((rc = sdrx_add_vfo(M.c, &d, &lid)) != SDRX_OK)
                return member_fail(g, k, rc);
            local[(size_t)i] = lid;
            M.global_of.push_back(i);
            if (g->where[(size_t)i].first < 0)
                g->where[(size_t)i] = std::make_pair(k, lid);
        }
        if ((rc = sdrx_finalize(M.c)) != SDRX_OK)
            return member_fail(g, k, rc);
        if (M.c->root_frame != g->root_frame)
            return gfail(g, SDRX_EINVAL, "member %d was initialised for frames of %d samples, the group for %d", k, M.c->root_frame, g->root_frame);
        if (k > 0)
            for (int p = 0; p < 2; ++p)
                GHIP(g, hipMalloc(&M.d_frame[p], (size_t)g->root_frame * sizeof(float2)));
    }
    // reference publish order over the WHOLE tree: main order x sub order (sdrj.cpp:288-294, vfo.cpp:257-263)
    std::vector<int> stack;
    for (auto it = roots.rbegin(); it != roots.rend(); ++it)
        stack.push_back(*it);
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        if (children[(size_t)i].empty())
            g->publish_order.push_back(i);
        else
            for (auto it = children[(size_t)i].rbegin(); it != children[(size_t)i].rend(); ++it)
                stack.push_back(*it);
    }
    GHIP(g, hipSetDevice(g->m[0].device));
    GHIP(g, hipStreamCreateWithFlags(&g->stage_stream, hipStreamNonBlocking));
    for (int p = 0; p < 2; ++p) {
        GHIP(g, hipHostMalloc(&g->h_stage[p], (size_t)g->root_frame * sizeof(float2), hipHostMallocDefault));
        GHIP(g, hipMalloc(&g->d_stage[p], (size_t)g->root_frame * sizeof(float2)));
        GHIP(g, hipEventCreateWithFlags(&g->ev_ready[p], hipEventDisableTiming));
    }
    g->finalized = true;
    return SDRX_OK;
}

int sdrx_group_submit(sdrx_group *g, const float *iq, int n_complex)
{
    int rc = group_check(g, "sdrx_group_submit", iq, n_complex, false);
    if (rc)
        return rc;
    const size_t bytes = (size_t)n_complex * sizeof(float2);
    rc = group_stage(g, iq, bytes);
    return rc ? rc : group_enqueue(g, g->d_stage[g->frame_no & 1ull], bytes, kRawF32, true);
}

// dongle bytes (jonti/sdr.cpp:43-49): a quarter of the bytes cross PCIe and xGMI; every device applies
// the b - 127 LUT itself and, with correct_dc, the DC-bias IIR of sdrj.cpp:271-286 with an accumulator of
// its own (the recurrence is sequential over the raw stream, so each member runs it on the whole frame: same
// bytes, same start state, same result on every device -- nothing but the bytes is exchanged).
int sdrx_group_submit_u8(sdrx_group *g, const uint8_t *bytes, int n_complex, int correct_dc)
{
    int rc = group_check(g, "sdrx_group_submit_u8", bytes, n_complex, false);
    if (rc)
        return rc;
    const size_t nb = (size_t)n_complex * 2;
    rc = group_stage(g, bytes, nb);
    return rc ? rc : group_enqueue(g, g->d_stage[g->frame_no & 1ull], nb, kRawU8, true, correct_dc);
}

int sdrx_group_process_u8(sdrx_group *g, const uint8_t *bytes, int n_complex, int correct_dc)
{
    int rc = group_check(g, "sdrx_group_process_u8", bytes, n_complex, true);
    if (rc)
        return rc;
    rc = sdrx_group_submit_u8(g, bytes, n_complex, correct_dc);
    return rc ? rc : sdrx_group_wait(g);
}

// 1: every member reaches the first device's frame directly (same device, or peer access enabled: one xGMI
// link per peer); 0: at least one peer copy is staged through host memory by the runtime.
int sdrx_group_peer_access(const sdrx_group *g) { return g ? (g->peer_ok ? 1 : 0) : SDRX_EINVAL; }

// `dev_iq`: n_complex cf32 on the FIRST device of the group, complete in the order of `producer_stream`
// (a hipStream_t of that device; NULL: complete already) at the time of the call; it must stay untouched
// until this frame has been waited for.
int sdrx_group_submit_device(sdrx_group *g, const void *dev_iq, int n_complex, void *producer_stream)
{
    int rc = group_check(g, "sdrx_group_submit_device", dev_iq, n_complex, false);
    if (rc)
        return rc;
    GHIP(g, hipSetDevice(g->m[0].device));
    GHIP(g, hipEventRecord(g->ev_ready[g->frame_no & 1ull], producer_stream ? reinterpret_cast<hipStream_t>(producer_stream) : g->stage_stream));
    return group_enqueue(g, dev_iq, (size_t)n_complex * sizeof(float2), kRawF32, true);
}

// The device-resident frame without the payload copies (cf. sdrx_process_device): every member queues its
// kernels and returns; sdrx_group_sync waits for all of them, sdrx_group_get_output fetches on demand.
int sdrx_group_process_device(sdrx_group *g, const void *dev_iq, int n_complex, void *producer_stream)
{
    int rc = group_check(g, "sdrx_group_process_device", dev_iq, n_complex, true);
    if (rc)
        return rc;
    GHIP(g, hipSetDevice(g->m[0].device));
    GHIP(g, hipEventRecord(g->ev_ready[g->frame_no & 1ull], producer_stream ? reinterpret_cast<hipStream_t>(producer_stream) : g->stage_stream));
    return group_enqueue(g, dev_iq, (size_t)n_complex * sizeof(float2), kRawF32, false);
}

int sdrx_group_wait(sdrx_group *g)
{
    if (!g)
        return SDRX_EINVAL;
    if (g->in_flight <= 0)
        return gfail(g, SDRX_ESTATE, "sdrx_group_wait: no submitted frame is in flight");
    for (size_t k = 0; k < g->m.size(); ++k) // (copies that are sdrx_wait's to issue: all devices' first, then the waits)
        if (g->m[k].c) {
            const int rc = start_owed_copy(g->m[k].c);
            if (rc) {
                g->broken = true;
                return member_fail(g, (int)k, rc);
            }
        }
    for (size_t k = 0; k < g->m.size(); ++k) {
        if (!g->m[k].c)
            continue;
        int slot = 0;
        const int rc = wait_frame(g->m[k].c, &slot);
        if (rc) {
            g->broken = true;
            return member_fail(g, (int)k, rc);
        }
    }
    g->in_flight--;
    group_publish(g);
    return SDRX_OK;
}

int sdrx_group_in_flight(sdrx_group *g) { return g ? g->in_flight : SDRX_EINVAL; }

int sdrx_group_process(sdrx_group *g, const float *iq, int n_complex)
{
    int rc = group_check(g, "sdrx_group_process", iq, n_complex, true);
    if (rc)
        return rc;
    rc = sdrx_group_submit(g, iq, n_complex);
    return rc ? rc : sdrx_group_wait(g);
}

int sdrx_group_sync(sdrx_group *g)
{
    if (!g)
        return SDRX_EINVAL;
    for (size_t k = 0; k < g->m.size(); ++k)
        if (g->m[k].c) {
            const int rc = sdrx_sync(g->m[k].c);
            if (rc)
                return member_fail(g, (int)k, rc);
        }
    return SDRX_OK;
}

// Payload of leaf `id` (an id of sdrx_group_add_vfo) of the last delivered frame.
int sdrx_group_get_output(sdrx_group *g, int id, const void **buf, uint32_t *len_bytes, uint32_t *rate)
{
    if (!g || id < 0 || id >= (int)g->descs.size())
        return gfail(g, SDRX_EINVAL, "bad vfo id %d", id);
    if (!g->finalized)
        return gfail(g, SDRX_ESTATE, "sdrx_group_get_output before sdrx_group_finalize");
    // (with frames in flight this keeps serving the last DELIVERED frame, like sdrx_get_output: its payloads sit in
    // the host slot the frame in flight does not write; before the first delivery the member says SDRX_ESTATE)
    const auto w = g->where[(size_t)id];
    if (w.first < 0)
        return gfail(g, SDRX_EINVAL, "vfo %d is held by no device", id);
    const int rc = sdrx_get_output(g->m[(size_t)w.first].c, w.second, buf, len_bytes, rate);
    return rc ? member_fail(g, w.first, rc) : SDRX_OK;
}

// Which member holds VFO `id` -- the owner of a leaf or sub VFO, the first replica of a VFO with
// children -- and its LOCAL id inside that member's context (sdrx_group_member), for sdrx_get_stream,
// sdrx_get_stats, kernel timing ...
int sdrx_group_locate(sdrx_group *g, int id, int *member, int *local_id)
{
    if (!g || id < 0 || id >= (int)g->descs.size())
        return gfail(g, SDRX_EINVAL, "bad vfo id %d", id);
    if (member)
        *member = g->where[(size_t)id].first;
    if (local_id)
        *local_id = g->where[(size_t)id].second;
    return SDRX_OK;
}

int sdrx_group_member(sdrx_group *g, int k, sdrx_ctx **ctx, int *device)
{
    if (!g || k < 0 || k >= (int)g->m.size())
        return gfail(g, SDRX_EINVAL, "bad member index %d", k);
    if (ctx)
        *ctx = g->m[(size_t)k].c; // NULL for a member that holds no VFOs
    if (device)
        *device = g->m[(size_t)k].device;
    return SDRX_OK;
}

} // extern "C"
