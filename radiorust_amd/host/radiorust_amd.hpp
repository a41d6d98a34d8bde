// radiorust_amd.hpp — C++17 host layer above the C ABI (include/radiorust_amd.h).
//
// The reference is compiled code (Rust) whose toolchain is absent here, so the
// host side that mirrors its block API is written in C++: same type and method
// names, same argument meaning, same message flow, one OS thread per block where
// the reference has one tokio task per block.  All arithmetic happens in the HIP
// library; this header moves messages, evaluates user closures and owns pinned
// buffers.  Citations are relative to /root/reference.
//
//   numbers      Complex<Flt>                          src/numbers.rs:10
//   bufferpool   Chunk, ChunkBuf, ChunkBufPool         src/bufferpool.rs:44-222 (pinned HIP host memory)
//   signal       Event, Disconnection, Signal          src/signal.rs:19-46,170-215
//   flow         new_sender/new_receiver, Producer/... src/flow.rs:103-273, src/sync/broadcast_bp.rs
//   windowing    Window, Rectangular, Kaiser, Custom   src/windowing.rs:6-67
//   blocks       FreqShifter, Filter, Downsampler, Fourier
//                src/blocks/{transform,filters,resampling,analysis}.rs
#pragma once

#include <atomic>
#include <chrono>
#include <complex>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/radiorust_amd.h"

namespace radiorust {

template <class Flt> using Complex = std::complex<Flt>;  // {re, im}, same layout as rr_c32 / rr_c64

struct BackendError : std::runtime_error {
    int status;
    BackendError(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};
// RR_ERR_CONTRACT: the reference would panic!
struct ContractViolation : BackendError {
    using BackendError::BackendError;
};
inline void check(int status) {
    if (status == RR_OK) return;
    const std::string msg = rr_last_error_string();
    if (status == RR_ERR_CONTRACT) throw ContractViolation(status, msg);
    throw BackendError(status, msg);
}
template <class Flt> constexpr int dtype_of() { return sizeof(Flt) == 4 ? RR_F32 : RR_F64; }

// ===========================================================================
// bufferpool — src/bufferpool.rs, re-backed by pinned HIP host memory
// ===========================================================================
namespace bufferpool {

template <class T> struct PoolCore;

template <class T> struct Storage {  // one pinned allocation (the reference's Vec<T>)
    T *ptr = nullptr;
    size_t len = 0, cap = 0;
    std::weak_ptr<PoolCore<T>> recycler;
    Storage() = default;
    Storage(const Storage &) = delete;
    Storage &operator=(const Storage &) = delete;
    ~Storage();
    void reserve(size_t want) {
        if (want <= cap) return;
        void *p = nullptr;
        check(rr_host_alloc(want * sizeof(T), &p));
        if (ptr) {
            std::memcpy(p, ptr, len * sizeof(T));
            rr_host_free(ptr);
        }
        ptr = static_cast<T *>(p);
        cap = want;
    }
};

template <class T> struct PoolCore {  // the mpsc channel of spare buffers (bufferpool.rs:187-190)
    std::mutex m;
    std::vector<Storage<T> *> spare;
    ~PoolCore() {
        for (Storage<T> *s : spare) {
            s->recycler.reset();
            delete s;
        }
    }
};

template <class T> Storage<T>::~Storage() {
    if (ptr) rr_host_free(ptr);
}

// last reference gone: hand the allocation back to its pool (bufferpool.rs:82-90)
template <class T> struct Recycle {
    void operator()(Storage<T> *s) const {
        if (auto pool = s->recycler.lock()) {
            std::lock_guard<std::mutex> g(pool->m);
            pool->spare.push_back(s);
        } else {
            delete s;
        }
    }
};

// Immutable, reference-counted view of samples (bufferpool.rs:44-48)
template <class T> class Chunk {
    std::shared_ptr<Storage<T>> buffer_;
    size_t begin_ = 0, end_ = 0;

  public:
    Chunk() = default;
    explicit Chunk(std::shared_ptr<Storage<T>> b) : buffer_(std::move(b)), begin_(0), end_(buffer_ ? buffer_->len : 0) {}
    static Chunk from(const std::vector<T> &v) {  // `Chunk::from(vec![..])`
        auto *s = new Storage<T>;
        s->reserve(v.size() ? v.size() : 1);
        std::memcpy(s->ptr, v.data(), v.size() * sizeof(T));
        s->len = v.size();
        return Chunk(std::shared_ptr<Storage<T>>(s, Recycle<T>{}));
    }
    size_t len() const { return end_ - begin_; }
    size_t size() const { return len(); }
    const T *data() const { return buffer_ ? buffer_->ptr + begin_ : nullptr; }
    const T &operator[](size_t i) const { return data()[i]; }
    const T *begin() const { return data(); }
    const T *end() const { return data() + len(); }
    void discard_beginning(size_t n) {  // bufferpool.rs:66-69
        if (n > len()) throw std::logic_error("length exceeded");
        begin_ += n;
    }
    Chunk separate_beginning(size_t n) {  // bufferpool.rs:70-79 (zero copy)
        if (n > len()) throw std::logic_error("length exceeded");
        Chunk head = *this;
        head.end_ = begin_ + n;
        begin_ += n;
        return head;
    }
};

// Writable buffer (bufferpool.rs:125-165)
template <class T> class ChunkBuf {
    Storage<T> *s_ = nullptr;

  public:
    explicit ChunkBuf(Storage<T> *s) : s_(s) {}
    ChunkBuf(ChunkBuf &&o) noexcept : s_(o.s_) { o.s_ = nullptr; }
    ChunkBuf(const ChunkBuf &) = delete;
    ~ChunkBuf() {
        if (s_) Recycle<T>{}(s_);
    }
    size_t len() const { return s_->len; }
    size_t capacity() const { return s_->cap; }
    T *data() { return s_->ptr; }
    void reserve(size_t n) { s_->reserve(n); }
    void resize(size_t n) {
        s_->reserve(n);
        s_->len = n;
    }
    void push(const T &v) {
        if (s_->len == s_->cap) s_->reserve(s_->cap ? 2 * s_->cap : 16);
        s_->ptr[s_->len++] = v;
    }
    void extend_from_slice(const T *p, size_t n) {
        s_->reserve(s_->len + n);
        std::memcpy(s_->ptr + s_->len, p, n * sizeof(T));
        s_->len += n;
    }
    void truncate(size_t n) {
        if (n < s_->len) s_->len = n;
    }
    Chunk<T> finalize() {  // bufferpool.rs:141-143
        Storage<T> *s = s_;
        s_ = nullptr;
        return Chunk<T>(std::shared_ptr<Storage<T>>(s, Recycle<T>{}));
    }
};

// Pool of pinned buffers (bufferpool.rs:187-222)
template <class T> class ChunkBufPool {
    std::shared_ptr<PoolCore<T>> core_ = std::make_shared<PoolCore<T>>();

  public:
    ChunkBuf<T> get_with_capacity(size_t capacity) {
        Storage<T> *s = nullptr;
        {
            std::lock_guard<std::mutex> g(core_->m);
            if (!core_->spare.empty()) {
                s = core_->spare.back();
                core_->spare.pop_back();
            }
        }
        if (s) {
            s->len = 0;  // `buffer.clear()`; a recycled buffer keeps its capacity (bufferpool.rs:214-218)
        } else {
            s = new Storage<T>;
            s->recycler = core_;
            s->reserve(capacity ? capacity : 1);
        }
        return ChunkBuf<T>(s);
    }
    ChunkBuf<T> get() { return get_with_capacity(0); }
    size_t spare_count() const {
        std::lock_guard<std::mutex> g(core_->m);
        return core_->spare.size();
    }
};

}  // namespace bufferpool
using bufferpool::Chunk;
using bufferpool::ChunkBuf;
using bufferpool::ChunkBufPool;

// ===========================================================================
// signal — src/signal.rs
// ===========================================================================
namespace signal {

struct Event {  // signal.rs:19-31
    virtual ~Event() = default;
    virtual bool is_interrupt() const { return false; }
    virtual bool is_flush() const { return false; }
    virtual const char *name() const { return "Event"; }
};
struct Disconnection : Event {  // signal.rs:37-46
    bool is_interrupt() const override { return true; }
    const char *name() const override { return "Disconnection"; }
};
struct SamplesLost : Event {  // blocks/chunks.rs:20-28
    bool is_interrupt() const override { return true; }
    const char *name() const override { return "SamplesLost"; }
};

// enum Signal<T> { Samples { sample_rate, chunk }, Event(Arc<dyn Event>) }  (signal.rs:170-183)
template <class T> struct Signal {
    double sample_rate = 0.0;
    Chunk<T> chunk;
    std::shared_ptr<const Event> event;  // non-null = Signal::Event

    static Signal Samples(double rate, Chunk<T> c) {
        Signal s;
        s.sample_rate = rate;
        s.chunk = std::move(c);
        return s;
    }
    template <class E> static Signal new_event(E e) {
        Signal s;
        s.event = std::make_shared<E>(std::move(e));
        return s;
    }
    static Signal Event_(std::shared_ptr<const Event> e) {
        Signal s;
        s.event = std::move(e);
        return s;
    }
    bool is_event() const { return event != nullptr; }
    double duration() const { return is_event() ? 0.0 : chunk.len() / sample_rate; }
    static std::optional<Signal> disconnection() { return new_event(Disconnection{}); }  // signal.rs:208-215
};

}  // namespace signal
using signal::Signal;

// ===========================================================================
// flow — src/sync/broadcast_bp.rs + src/flow.rs (threads instead of tokio tasks)
// ===========================================================================
namespace flow {

// `trait Message { fn disconnection() -> Option<Self> }` (flow.rs:74-78): plain
// values have none, Signal<T> yields a Disconnection event (signal.rs:208-215)
template <class T> struct MessageTraits {
    static std::optional<T> disconnection() { return std::nullopt; }
};
template <class U> struct MessageTraits<signal::Signal<U>> {
    static std::optional<signal::Signal<U>> disconnection() { return signal::Signal<U>::disconnection(); }
};

template <class T> struct Shared {  // broadcast_bp.rs:72-100
    std::mutex m;
    std::condition_variable notify_sndr, notify_rcvr;
    std::optional<T> data;
    int slot = 0;
    size_t sndr_count = 1, elst_count = 1, rcvr_count = 0, unseen = 0;
};

enum class RecvStatus { Ok, Closed, Timeout };

template <class T> class InnerReceiver {  // broadcast_bp::Receiver
    std::shared_ptr<Shared<T>> sh_;
    int slot_ = 0;

  public:
    InnerReceiver(std::shared_ptr<Shared<T>> sh, int slot) : sh_(std::move(sh)), slot_(slot) {}
    InnerReceiver(const InnerReceiver &) = delete;
    InnerReceiver(InnerReceiver &&o) noexcept : sh_(std::move(o.sh_)), slot_(o.slot_) {}
    ~InnerReceiver() {  // broadcast_bp.rs:171-187
        if (!sh_) return;
        std::lock_guard<std::mutex> g(sh_->m);
        sh_->rcvr_count -= 1;
        bool notify = sh_->rcvr_count == 0 && sh_->elst_count == 0;
        if (slot_ != sh_->slot) {
            sh_->unseen -= 1;
            if (sh_->unseen == 0) notify = true;
        }
        if (notify) sh_->notify_sndr.notify_all();
    }
    // broadcast_bp.rs:309-331: the last reader takes the value, the others clone it
    RecvStatus recv(T &out, std::chrono::milliseconds timeout) {
        std::unique_lock<std::mutex> g(sh_->m);
        const auto deadline = std::chrono::steady_clock::now() + timeout;
        while (true) {
            if (slot_ != sh_->slot) {
                slot_ = sh_->slot;
                sh_->unseen -= 1;
                if (sh_->unseen == 0) {
                    out = std::move(*sh_->data);
                    sh_->data.reset();
                    sh_->notify_sndr.notify_all();
                } else {
                    out = *sh_->data;
                }
                return RecvStatus::Ok;
            }
            if (sh_->sndr_count == 0) return RecvStatus::Closed;
            if (sh_->notify_rcvr.wait_until(g, deadline) == std::cv_status::timeout && slot_ == sh_->slot)
                return sh_->sndr_count == 0 ? RecvStatus::Closed : RecvStatus::Timeout;
        }
    }
};

template <class T> class SenderConnector {  // broadcast_bp::Enlister
    std::shared_ptr<Shared<T>> sh_;

  public:
    explicit SenderConnector(std::shared_ptr<Shared<T>> sh) : sh_(std::move(sh)) {}
    SenderConnector(const SenderConnector &o) : sh_(o.sh_) {
        std::lock_guard<std::mutex> g(sh_->m);
        sh_->elst_count += 1;
    }
    SenderConnector &operator=(const SenderConnector &) = delete;
    ~SenderConnector() {
        std::lock_guard<std::mutex> g(sh_->m);
        sh_->elst_count -= 1;
        if (sh_->elst_count == 0 && sh_->rcvr_count == 0) sh_->notify_sndr.notify_all();
    }
    InnerReceiver<T> subscribe() const {  // broadcast_bp.rs:103-113
        std::lock_guard<std::mutex> g(sh_->m);
        sh_->rcvr_count += 1;
        sh_->notify_sndr.notify_all();
        return InnerReceiver<T>(sh_, sh_->slot);
    }
    const SenderConnector &sender_connector() const { return *this; }  // impl Producer for SenderConnector
};

template <class T> class Sender {  // broadcast_bp::Sender (capacity 1, back-pressure)
    std::shared_ptr<Shared<T>> sh_;

  public:
    explicit Sender(std::shared_ptr<Shared<T>> sh) : sh_(std::move(sh)) {}
    Sender(Sender &&o) noexcept : sh_(std::move(o.sh_)) {}
    Sender(const Sender &) = delete;
    ~Sender() {
        if (!sh_) return;
        std::lock_guard<std::mutex> g(sh_->m);
        sh_->sndr_count -= 1;
        if (sh_->sndr_count == 0) sh_->notify_rcvr.notify_all();
    }
    // Not in the reference: blocks until `n` receivers have subscribed (or `timeout_ms` have passed).  A value is
    // delivered to the receivers subscribed at the time of the send (broadcast_bp.rs:273-291); under tokio's test
    // runtime the reference's tests have every consumer task subscribed before their first send runs, with one OS
    // thread per block that order has to be asked for where ONE sender feeds several blocks.
    bool wait_for_receivers(size_t n, int timeout_ms = 10000) {
        std::unique_lock<std::mutex> g(sh_->m);
        const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
        while (sh_->rcvr_count < n) {
            if (std::chrono::steady_clock::now() >= until) return false;
            sh_->notify_sndr.wait_for(g, std::chrono::milliseconds(1));
        }
        return true;
    }
    // broadcast_bp.rs:230-291: waits until every receiver has seen the previous
    // value and at least one receiver exists; false = nobody can ever receive
    bool send(T msg, const std::atomic<bool> *stop = nullptr) {
        std::unique_lock<std::mutex> g(sh_->m);
        while (true) {
            if (sh_->rcvr_count == 0 && sh_->elst_count == 0) return false;
            if (sh_->unseen == 0 && sh_->rcvr_count > 0) break;
            if (stop && stop->load()) return false;
            sh_->notify_sndr.wait_for(g, std::chrono::milliseconds(2));
        }
        sh_->slot ^= 1;
        sh_->data = std::move(msg);
        sh_->unseen = sh_->rcvr_count;
        sh_->notify_rcvr.notify_all();
        return true;
    }
};

template <class T> std::pair<Sender<T>, SenderConnector<T>> new_sender() {  // broadcast_bp::channel
    auto sh = std::make_shared<Shared<T>>();
    return {Sender<T>(sh), SenderConnector<T>(sh)};
}

// watch::channel(Option<Enlister>) of flow.rs:103-152
template <class T> struct ConnState {
    std::mutex m;
    std::condition_variable cv;
    std::unique_ptr<SenderConnector<T>> enlister;
    uint64_t version = 0;
    bool tx_alive = true;
};

template <class T> class Receiver {  // flow.rs:113-226
    std::shared_ptr<ConnState<T>> st_;
    uint64_t seen_ = 0;
    std::optional<InnerReceiver<T>> inner_;

    std::optional<T> change() {  // the `change` closure of flow.rs:177-189
        const bool was_connected = inner_.has_value();
        inner_.reset();
        {
            std::lock_guard<std::mutex> g(st_->m);
            seen_ = st_->version;
            if (st_->enlister) inner_.emplace(st_->enlister->subscribe());
        }
        if (was_connected) return MessageTraits<T>::disconnection();
        return std::nullopt;
    }

  public:
    explicit Receiver(std::shared_ptr<ConnState<T>> st) : st_(std::move(st)) {
        std::lock_guard<std::mutex> g(st_->m);
        seen_ = st_->version;
        if (st_->enlister) inner_.emplace(st_->enlister->subscribe());
    }
    Receiver(Receiver &&) = default;
    // Ok(message) or nullopt = RecvError (no sender and no way to get one) / stop requested
    std::optional<T> recv(const std::atomic<bool> *stop = nullptr) {
        while (true) {
            if (stop && stop->load()) return std::nullopt;
            bool changed, tx_alive;
            {
                std::lock_guard<std::mutex> g(st_->m);
                changed = st_->version != seen_;
                tx_alive = st_->tx_alive;
            }
            if (changed) {
                if (auto m = change()) return m;
                continue;
            }
            if (inner_) {
                T out;
                switch (inner_->recv(out, std::chrono::milliseconds(2))) {
                    case RecvStatus::Ok: return out;
                    case RecvStatus::Closed: inner_.reset(); break;
                    case RecvStatus::Timeout: break;
                }
            } else {
                if (!tx_alive) return std::nullopt;
                std::unique_lock<std::mutex> g(st_->m);
                st_->cv.wait_for(g, std::chrono::milliseconds(2));
            }
        }
    }
};

template <class T> class ReceiverConnector {  // flow.rs:103-152
    std::shared_ptr<ConnState<T>> st_ = std::make_shared<ConnState<T>>();

  public:
    ReceiverConnector() = default;
    ReceiverConnector(const ReceiverConnector &) = delete;
    ~ReceiverConnector() {
        std::lock_guard<std::mutex> g(st_->m);
        st_->tx_alive = false;
        st_->cv.notify_all();
    }
    void connect(const SenderConnector<T> &c) {
        std::lock_guard<std::mutex> g(st_->m);
        st_->enlister = std::make_unique<SenderConnector<T>>(c);
        st_->version += 1;
        st_->cv.notify_all();
    }
    void disconnect() {
        std::lock_guard<std::mutex> g(st_->m);
        st_->enlister.reset();
        st_->version += 1;
        st_->cv.notify_all();
    }
    Receiver<T> stream() { return Receiver<T>(st_); }
    ReceiverConnector &receiver_connector() { return *this; }  // impl Consumer for ReceiverConnector
    template <class P> void feed_from(const P &producer) { connect(producer.sender_connector()); }
};

template <class T> std::pair<Receiver<T>, std::unique_ptr<ReceiverConnector<T>>> new_receiver() {  // flow.rs:132-136
    auto rc = std::make_unique<ReceiverConnector<T>>();
    Receiver<T> r = rc->stream();
    return {std::move(r), std::move(rc)};
}

// Producer / Consumer (flow.rs:233-267) as mix-ins over the two connectors
template <class Derived, class T> struct Producer {
    template <class C> void feed_into(C &consumer) const {
        consumer.receiver_connector().connect(static_cast<const Derived *>(this)->sender_connector());
    }
};
template <class Derived, class T> struct Consumer {
    template <class P> void feed_from(const P &producer) {
        static_cast<Derived *>(this)->receiver_connector().connect(producer.sender_connector());
    }
    void feed_from_none() { static_cast<Derived *>(this)->receiver_connector().disconnect(); }
};

}  // namespace flow
using flow::new_receiver;
using flow::new_sender;

// ===========================================================================
// windowing — src/windowing.rs (values from the library's own f64 design math)
// ===========================================================================
namespace windowing {

struct Window {  // windowing.rs:6-10
    virtual ~Window() = default;
    virtual double relative_value_at(double x) const = 0;
    virtual bool builtin(rr_window *spec) const {
        (void)spec;
        return false;
    }
    // values at 2 (i + 0.5) / n - 1  (filters.rs:209-212, analysis.rs:93-94)
    std::vector<double> sample(size_t n) const {
        std::vector<double> out(n);
        rr_window spec;
        if (builtin(&spec)) {
            check(rr_window_sample(&spec, n, out.data()));
        } else {
            for (size_t i = 0; i < n; ++i) out[i] = relative_value_at(2.0 * ((double)i + 0.5) / (double)n - 1.0);
        }
        return out;
    }
};
struct Rectangular : Window {
    double relative_value_at(double) const override { return 1.0; }
    bool builtin(rr_window *s) const override {
        *s = rr_window{RR_WIN_RECTANGULAR, 0.0};
        return true;
    }
};
struct Kaiser : Window {
    double beta;
    explicit Kaiser(double b) : beta(b) {}
    static Kaiser with_beta(double b) { return Kaiser(b); }
    static Kaiser with_alpha(double a) { return Kaiser(rr_kaiser_alpha_to_beta(a)); }
    static Kaiser with_null_at_bin(double n) { return Kaiser(rr_kaiser_null_at_bin_to_beta(n)); }
    double relative_value_at(double x) const override { return rr_kaiser_rel_with_beta(beta, x); }
    bool builtin(rr_window *s) const override {
        *s = rr_window{RR_WIN_KAISER, beta};
        return true;
    }
};
struct CustomWindow : Window {  // windowing.rs:58-67
    std::function<double(double)> f;
    explicit CustomWindow(std::function<double(double)> fn) : f(std::move(fn)) {}
    double relative_value_at(double x) const override { return f(x); }
};

}  // namespace windowing

// ===========================================================================
// blocks
// ===========================================================================
namespace blocks {

// The skeleton every block shares (blocks/mod.rs:193-239): a receiver, a sender and
// one worker that turns each received message into zero or more sent ones.
template <class Derived, class Flt>
class BlockBase : public flow::Producer<Derived, Signal<Complex<Flt>>>, public flow::Consumer<Derived, Signal<Complex<Flt>>> {
  public:
    using Sig = Signal<Complex<Flt>>;

  protected:
    std::unique_ptr<flow::ReceiverConnector<Sig>> receiver_connector_;
    std::unique_ptr<flow::SenderConnector<Sig>> sender_connector_;
    std::atomic<bool> stop_{false};
    std::thread worker_;

    // body(signal, send) handles one message; returning false ends the task
    template <class Body> void spawn(Body body) {
        auto rx = flow::new_receiver<Sig>();
        auto tx = flow::new_sender<Sig>();
        receiver_connector_ = std::move(rx.second);
        sender_connector_ = std::make_unique<flow::SenderConnector<Sig>>(tx.second);
        worker_ = std::thread([this, receiver = std::move(rx.first), sender = std::move(tx.first), body]() mutable {
            while (true) {
                std::optional<Sig> msg = receiver.recv(&stop_);
                if (!msg) return;  // `let Ok(signal) = receiver.recv().await else { return; }`
                auto send = [&](Sig s) { return sender.send(std::move(s), &stop_); };
                if (!body(std::move(*msg), send)) return;
            }
        });
    }
    void shutdown() {
        stop_.store(true);
        if (worker_.joinable()) worker_.join();
    }

  public:
    flow::ReceiverConnector<Sig> &receiver_connector() { return *receiver_connector_; }
    const flow::SenderConnector<Sig> &sender_connector() const { return *sender_connector_; }
};

// ---- FreqShifter (transform.rs:266-391) -------------------------------------------
template <class Flt> class FreqShifter : public BlockBase<FreqShifter<Flt>, Flt> {
    using Base = BlockBase<FreqShifter<Flt>, Flt>;
    using Sig = typename Base::Sig;
    rr_freqshifter *h_ = nullptr;
    double precision_;
    std::mutex m_;
    double shift_;
    bool shift_changed_ = false;

  public:
    FreqShifter() : FreqShifter(1.0, 0.0) {}
    FreqShifter(double precision, double shift, int device = 0) : precision_(precision), shift_(shift) {
        check(rr_freqshifter_create(dtype_of<Flt>(), precision, shift, device, &h_));
        auto pool = std::make_shared<ChunkBufPool<Complex<Flt>>>();
        this->spawn([this, pool](Sig signal, auto &send) {
            if (signal.is_event()) return send(std::move(signal));  // transform.rs:357-359
            {
                std::lock_guard<std::mutex> g(m_);
                if (shift_changed_) {
                    check(rr_freqshifter_set_shift(h_, shift_));
                    shift_changed_ = false;
                }
            }
            const size_t n = signal.chunk.len();
            auto out = pool->get_with_capacity(n);
            out.resize(n);
            size_t n_out = 0;
            check(rr_freqshifter_process(h_, signal.sample_rate, signal.chunk.data(), n, out.data(), n, &n_out));
            out.truncate(n_out);
            return send(Sig::Samples(signal.sample_rate, out.finalize()));
        });
    }
    ~FreqShifter() {
        this->shutdown();
        rr_freqshifter_destroy(h_);
    }
    static std::unique_ptr<FreqShifter> with_shift(double shift) { return std::make_unique<FreqShifter>(1.0, shift); }
    static std::unique_ptr<FreqShifter> with_precision(double p) { return std::make_unique<FreqShifter>(p, 0.0); }
    static std::unique_ptr<FreqShifter> with_precision_and_shift(double p, double s) { return std::make_unique<FreqShifter>(p, s); }
    double precision() const { return precision_; }
    double shift() {
        std::lock_guard<std::mutex> g(m_);
        return shift_;
    }
    void set_shift(double s) {  // transform.rs:384-386
        std::lock_guard<std::mutex> g(m_);
        shift_ = s;
        shift_changed_ = true;
    }
    template <class F> void update_shift(F modify) {
        std::lock_guard<std::mutex> g(m_);
        modify(shift_);
        shift_changed_ = true;
    }
};

// ---- Filter (filters.rs:110-298) ------------------------------------------------------
template <class Flt> class Filter : public BlockBase<Filter<Flt>, Flt> {
    using Base = BlockBase<Filter<Flt>, Flt>;
    using Sig = typename Base::Sig;

  public:
    using FreqResp = std::function<Complex<double>(long bin, double freq)>;

  private:
    rr_filter *h_ = nullptr;
    std::mutex m_;
    FreqResp freq_resp_;
    std::shared_ptr<const windowing::Window> window_;
    bool params_changed_ = false;

    void ensure_design(double rate, size_t n) {
        FreqResp fr;
        std::shared_ptr<const windowing::Window> win;
        {
            std::lock_guard<std::mutex> g(m_);
            if (params_changed_) {
                check(rr_filter_mark_params_changed(h_));
                params_changed_ = false;
            }
            fr = freq_resp_;
            win = window_;
        }
        int needed = 0;
        check(rr_filter_needs_design(h_, rate, n, &needed));
        if (!needed) return;
        std::vector<rr_c64> resp(n, rr_c64{0.0, 0.0});  // filters.rs:188-199
        const double step = rate / (double)n;
        for (size_t i = 0; n && i <= (n - 1) / 2; ++i) {
            const Complex<double> v = fr((long)i, (double)i * step);
            resp[i] = rr_c64{v.real(), v.imag()};
            if (i > 0) {
                const Complex<double> w = fr(-(long)i, -((double)i * step));
                resp[n - i] = rr_c64{w.real(), w.imag()};
            }
        }
        const std::vector<double> wv = win->sample(n);
        check(rr_filter_design(h_, rate, n, resp.data(), wv.data()));
    }

  public:
    Filter(FreqResp freq_resp, std::shared_ptr<const windowing::Window> window, int device = 0)
        : freq_resp_(std::move(freq_resp)), window_(std::move(window)) {
        check(rr_filter_create(dtype_of<Flt>(), device, &h_));
        auto pool = std::make_shared<ChunkBufPool<Complex<Flt>>>();
        this->spawn([this, pool](Sig signal, auto &send) {
            if (signal.is_event()) {
                if (signal.event->is_interrupt()) check(rr_filter_reset(h_));  // filters.rs:262-265
                return send(std::move(signal));
            }
            const size_t n = signal.chunk.len();
            ensure_design(signal.sample_rate, n);
            auto out = pool->get_with_capacity(n);
            out.resize(n);
            size_t n_out = 0;
            check(rr_filter_process(h_, signal.sample_rate, signal.chunk.data(), n, out.data(), n, &n_out));
            if (n_out == 0) return true;  // first chunk after a reset: nothing to send
            out.truncate(n_out);
            return send(Sig::Samples(signal.sample_rate, out.finalize()));
        });
    }
    ~Filter() {
        this->shutdown();
        rr_filter_destroy(h_);
    }
    // filters.rs:128-152
    static std::unique_ptr<Filter> new_(FreqResp f) {
        return std::make_unique<Filter>(std::move(f), std::make_shared<windowing::Kaiser>(windowing::Kaiser::with_null_at_bin(2.0)));
    }
    static std::unique_ptr<Filter> new_rectangular(FreqResp f) {
        return std::make_unique<Filter>(std::move(f), std::make_shared<windowing::Rectangular>());
    }
    template <class W> static std::unique_ptr<Filter> with_window(FreqResp f, W window) {
        return std::make_unique<Filter>(std::move(f), std::make_shared<W>(std::move(window)));
    }
    void update(FreqResp f) {  // filters.rs:279-287
        std::lock_guard<std::mutex> g(m_);
        freq_resp_ = std::move(f);
        params_changed_ = true;
    }
    template <class W> void update_with_window(FreqResp f, W window) {  // filters.rs:288-297
        std::lock_guard<std::mutex> g(m_);
        freq_resp_ = std::move(f);
        window_ = std::make_shared<W>(std::move(window));
        params_changed_ = true;
    }
};

// ---- Downsampler (resampling.rs:14-146) -----------------------------------------------
template <class Flt> class Downsampler : public BlockBase<Downsampler<Flt>, Flt> {
    using Base = BlockBase<Downsampler<Flt>, Flt>;
    using Sig = typename Base::Sig;
    rr_downsampler *h_ = nullptr;

  public:
    Downsampler(size_t output_chunk_len, double output_rate, double bandwidth, double quality = 3.0, int device = 0) {
        check(rr_downsampler_create(dtype_of<Flt>(), output_rate, bandwidth, quality, device, &h_));  // asserts of :51-56
        auto pool = std::make_shared<ChunkBufPool<Complex<Flt>>>();
        auto scratch = std::make_shared<ChunkBufPool<Complex<Flt>>>();
        auto output_chunk = std::make_shared<std::optional<ChunkBuf<Complex<Flt>>>>();
        output_chunk->emplace(pool->get_with_capacity(output_chunk_len));
        this->spawn([this, pool, scratch, output_chunk, output_chunk_len, output_rate](Sig signal, auto &send) {
            if (signal.is_event()) return send(std::move(signal));  // resampling.rs:135-137 (no reset)
            const size_t n = signal.chunk.len();
            size_t produce = 0;
            check(rr_downsampler_peek(h_, signal.sample_rate, n, &produce));
            auto raw = scratch->get_with_capacity(produce ? produce : 1);
            raw.resize(produce);
            size_t n_out = 0;
            check(rr_downsampler_process(h_, signal.sample_rate, signal.chunk.data(), n, raw.data(), produce, &n_out));
            // regroup into chunks of output_chunk_len (resampling.rs:121-131)
            for (size_t i = 0; i < n_out; ++i) {
                (*output_chunk)->push(raw.data()[i]);
                if ((*output_chunk)->len() >= output_chunk_len) {
                    Chunk<Complex<Flt>> done = (*output_chunk)->finalize();
                    output_chunk->emplace(pool->get_with_capacity(output_chunk_len));
                    if (!send(Sig::Samples(output_rate, std::move(done)))) return false;
                }
            }
            return true;
        });
    }
    ~Downsampler() {
        this->shutdown();
        rr_downsampler_destroy(h_);
    }
    static std::unique_ptr<Downsampler> new_(size_t output_chunk_len, double output_rate, double bandwidth) {
        return std::make_unique<Downsampler>(output_chunk_len, output_rate, bandwidth, 3.0);
    }
    static std::unique_ptr<Downsampler> with_quality(size_t l, double r, double b, double q) {
        return std::make_unique<Downsampler>(l, r, b, q);
    }
};

// ---- Upsampler (resampling.rs:147-280) ----------------------------------------------------
template <class Flt> class Upsampler : public BlockBase<Upsampler<Flt>, Flt> {
    using Base = BlockBase<Upsampler<Flt>, Flt>;
    using Sig = typename Base::Sig;
    rr_upsampler *h_ = nullptr;

  public:
    Upsampler(size_t output_chunk_len, double output_rate, double bandwidth, double quality = 3.0, int device = 0) {
        check(rr_upsampler_create(dtype_of<Flt>(), output_rate, bandwidth, quality, device, &h_));  // asserts of :185-186
        auto pool = std::make_shared<ChunkBufPool<Complex<Flt>>>();
        auto scratch = std::make_shared<ChunkBufPool<Complex<Flt>>>();
        auto output_chunk = std::make_shared<std::optional<ChunkBuf<Complex<Flt>>>>();
        output_chunk->emplace(pool->get_with_capacity(output_chunk_len));
        this->spawn([this, pool, scratch, output_chunk, output_chunk_len, output_rate](Sig signal, auto &send) {
            if (signal.is_event()) return send(std::move(signal));  // resampling.rs:269-271
            const size_t n = signal.chunk.len();
            size_t produce = 0;
            check(rr_upsampler_peek(h_, signal.sample_rate, n, &produce));
            auto raw = scratch->get_with_capacity(produce ? produce : 1);
            raw.resize(produce);
            size_t n_out = 0;
            check(rr_upsampler_process(h_, signal.sample_rate, signal.chunk.data(), n, raw.data(), produce, &n_out));
            for (size_t i = 0; i < n_out; ++i) {  // regroup into chunks of output_chunk_len (resampling.rs:251-261)
                (*output_chunk)->push(raw.data()[i]);
                if ((*output_chunk)->len() >= output_chunk_len) {
                    Chunk<Complex<Flt>> done = (*output_chunk)->finalize();
                    output_chunk->emplace(pool->get_with_capacity(output_chunk_len));
                    if (!send(Sig::Samples(output_rate, std::move(done)))) return false;
                }
            }
            return true;
        });
    }
    ~Upsampler() {
        this->shutdown();
        rr_upsampler_destroy(h_);
    }
    static std::unique_ptr<Upsampler> new_(size_t output_chunk_len, double output_rate, double bandwidth) {
        return std::make_unique<Upsampler>(output_chunk_len, output_rate, bandwidth, 3.0);
    }
    static std::unique_ptr<Upsampler> with_quality(size_t l, double r, double b, double q) {
        return std::make_unique<Upsampler>(l, r, b, q);
    }
};

// ---- FmDemod (modulation.rs:83-158) -------------------------------------------------------
template <class Flt> class FmDemod : public BlockBase<FmDemod<Flt>, Flt> {
    using Base = BlockBase<FmDemod<Flt>, Flt>;
    using Sig = typename Base::Sig;
    rr_fmdemod *h_ = nullptr;
    std::atomic<double> deviation_;  // the watch channel of modulation.rs:104

  public:
    explicit FmDemod(double deviation, int device = 0) : deviation_(deviation) {
        check(rr_fmdemod_create(dtype_of<Flt>(), deviation, device, &h_));
        auto pool = std::make_shared<ChunkBufPool<Complex<Flt>>>();
        this->spawn([this, pool](Sig signal, auto &send) {
            if (signal.is_event()) {
                if (signal.event->is_interrupt()) check(rr_fmdemod_reset(h_));  // modulation.rs:145-149
                return send(std::move(signal));
            }
            check(rr_fmdemod_set_deviation(h_, deviation_.load()));  // has_changed() / borrow_and_update()
            const size_t n = signal.chunk.len();
            auto out = pool->get_with_capacity(n ? n : 1);
            out.resize(n);
            size_t n_out = 0;
            check(rr_fmdemod_process(h_, signal.sample_rate, signal.chunk.data(), n, out.data(), n, &n_out));
            out.truncate(n_out);
            return send(Sig::Samples(signal.sample_rate, out.finalize()));
        });
    }
    ~FmDemod() {
        this->shutdown();
        rr_fmdemod_destroy(h_);
    }
    double deviation() const { return deviation_.load(); }                 // modulation.rs:163-165
    FmDemod &set_deviation(double d) {                                      // modulation.rs:167-170
        deviation_.store(d);
        return *this;
    }
};

// ---- Fourier (analysis.rs:26-133) ---------------------------------------------------------
template <class Flt> class Fourier : public BlockBase<Fourier<Flt>, Flt> {
    using Base = BlockBase<Fourier<Flt>, Flt>;
    using Sig = typename Base::Sig;
    rr_fourier *h_ = nullptr;
    std::shared_ptr<const windowing::Window> window_;
    bool sampled_ = false;
    size_t sampled_for_ = 0;

  public:
    Fourier(std::shared_ptr<const windowing::Window> window, bool center_dc, int device = 0) : window_(std::move(window)) {
        rr_window spec;
        if (!window_->builtin(&spec)) {
            spec = rr_window{RR_WIN_SAMPLED, 0.0};
            sampled_ = true;
        }
        check(rr_fourier_create(dtype_of<Flt>(), &spec, center_dc ? 1 : 0, device, &h_));
        auto pool = std::make_shared<ChunkBufPool<Complex<Flt>>>();
        this->spawn([this, pool](Sig signal, auto &send) {
            if (signal.is_event()) return send(std::move(signal));  // analysis.rs:122-124
            const size_t n = signal.chunk.len();
            if (sampled_ && sampled_for_ != n) {  // analysis.rs:82-104 for closures
                const std::vector<double> wv = window_->sample(n);
                check(rr_fourier_set_sampled_window(h_, n, wv.data()));
                sampled_for_ = n;
            }
            auto out = pool->get_with_capacity(n);
            out.resize(n);
            size_t n_out = 0;
            check(rr_fourier_process(h_, signal.chunk.data(), n, out.data(), n, &n_out));
            out.truncate(n_out);
            return send(Sig::Samples(signal.sample_rate, out.finalize()));
        });
    }
    ~Fourier() {
        this->shutdown();
        rr_fourier_destroy(h_);
    }
    // analysis.rs:39-59
    static std::unique_ptr<Fourier> new_() { return std::make_unique<Fourier>(std::make_shared<windowing::Rectangular>(), false); }
    static std::unique_ptr<Fourier> new_center_dc() { return std::make_unique<Fourier>(std::make_shared<windowing::Rectangular>(), true); }
    template <class W> static std::unique_ptr<Fourier> with_window(W w) {
        return std::make_unique<Fourier>(std::make_shared<W>(std::move(w)), false);
    }
    template <class W> static std::unique_ptr<Fourier> with_window_center_dc(W w) {
        return std::make_unique<Fourier>(std::make_shared<W>(std::move(w)), true);
    }
};

}  // namespace blocks
}  // namespace radiorust
