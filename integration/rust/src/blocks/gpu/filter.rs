//! [`Filter`] on the MI355X: drop-in for `blocks::filters::Filter` (`filters.rs:110-298`).

use super::bufferpool::PinnedChunkBufPool;
use super::{check, ffi, sample_window, GpuFloat, Handle};
use crate::flow::*;
use crate::impl_block_trait;
use crate::numbers::*;
use crate::signal::*;
use crate::windowing::{Kaiser, Rectangular, Window};

use tokio::sync::watch;
use tokio::task::spawn;

use std::os::raw::{c_int, c_void};
use std::ptr;

trait FreqRespFunc: Fn(isize, f64) -> Complex<f64> {}
impl<T: ?Sized> FreqRespFunc for T where T: Fn(isize, f64) -> Complex<f64> {}

struct FilterParams {
    freq_resp: Box<dyn FreqRespFunc + Send + Sync>,
    window: Box<dyn Window + Send + Sync>,
}

/// General purpose frequency filter using fast convolution (GPU version)
///
/// Same contract as the CPU block: the closure maps (DFT bin, signed frequency in hertz) to a complex
/// amplification factor; the impulse response is as long as the received chunks; the delay is one chunk (the
/// first chunk after start, redesign or an interrupting event produces no output).  The closure and the
/// window are evaluated on the host where the CPU block evaluates them (`filters.rs:188-199,209-212`); the
/// design itself (inverse transform, half swap, window, energy rescale, `filters.rs:200-225`) and the
/// convolution run behind `rr_filter_design` / `rr_filter_enqueue`.
pub struct Filter<Flt> {
    receiver_connector: ReceiverConnector<Signal<Complex<Flt>>>,
    sender_connector: SenderConnector<Signal<Complex<Flt>>>,
    params: watch::Sender<FilterParams>,
}

impl_block_trait! { <Flt> Consumer<Signal<Complex<Flt>>> for Filter<Flt> }
impl_block_trait! { <Flt> Producer<Signal<Complex<Flt>>> for Filter<Flt> }

/// `response[i]` as the CPU block fills it before dividing by `scale` (`filters.rs:188-199`)
fn sample_response(params: &FilterParams, n: usize, sample_rate: f64) -> Vec<ffi::rr_c64> {
    let mut response = vec![ffi::rr_c64 { re: 0.0, im: 0.0 }; n];
    let freq_step = sample_rate / n as f64;
    for i in 0..=(n - 1) / 2 {
        let freq = i as f64 * freq_step;
        let v = (params.freq_resp)(i as isize, freq);
        response[i] = ffi::rr_c64 { re: v.re, im: v.im };
        if i > 0 {
            let v = (params.freq_resp)(-(i as isize), -freq);
            response[n - i] = ffi::rr_c64 { re: v.re, im: v.im };
        }
    }
    response
}

impl<Flt> Filter<Flt>
where
    Flt: GpuFloat,
{
    /// Create new `Filter` block with given frequency response with Kaiser window
    /// ([`Kaiser::with_null_at_bin(2.0)`](Kaiser::with_null_at_bin))
    pub fn new<F>(freq_resp: F) -> Self
    where
        F: Fn(isize, f64) -> Complex<f64> + Send + Sync + 'static,
    {
        Self::new_internal(Box::new(freq_resp), Box::new(Kaiser::with_null_at_bin(2.0)))
    }
    /// Create new `Filter` block with given frequency response with rectangular window
    pub fn new_rectangular<F>(freq_resp: F) -> Self
    where
        F: Fn(isize, f64) -> Complex<f64> + Send + Sync + 'static,
    {
        Self::new_internal(Box::new(freq_resp), Box::new(Rectangular))
    }
    /// Create new `Filter` block with given frequency response and window function
    pub fn with_window<F, W>(freq_resp: F, window: W) -> Self
    where
        F: Fn(isize, f64) -> Complex<f64> + Send + Sync + 'static,
        W: Window + Send + Sync + 'static,
    {
        Self::new_internal(Box::new(freq_resp), Box::new(window))
    }
    fn new_internal(
        freq_resp: Box<dyn FreqRespFunc + Send + Sync>,
        window: Box<dyn Window + Send + Sync>,
    ) -> Self {
        let (mut receiver, receiver_connector) = new_receiver::<Signal<Complex<Flt>>>();
        let (sender, sender_connector) = new_sender::<Signal<Complex<Flt>>>();
        let (params_send, mut params_recv) = watch::channel(FilterParams { freq_resp, window });
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::rr_filter_create(Flt::DTYPE, 0, &mut raw) }).expect("radiorust_amd: no usable MI355X");
        let handle = Handle::new(raw, ffi::rr_filter_destroy);
        spawn(async move {
            let mut buf_pool = PinnedChunkBufPool::<Complex<Flt>>::new();
            loop {
                let Ok(signal) = receiver.recv().await else { return; };
                match signal {
                    Signal::Samples { sample_rate, chunk: input_chunk } => {
                        let n = input_chunk.len();
                        if n == 0 {
                            continue;
                        }
                        // `recalculate` (filters.rs:178-183): new parameters, sample rate or chunk length
                        if params_recv.has_changed().unwrap_or(false) {
                            unsafe { ffi::rr_filter_mark_params_changed(handle.get()) };
                        }
                        let mut needed: c_int = 0;
                        if check(unsafe { ffi::rr_filter_needs_design(handle.get(), sample_rate, n, &mut needed) }).is_err() {
                            return;
                        }
                        if needed != 0 {
                            let (response, window_rel) = {
                                let params = params_recv.borrow_and_update();
                                (sample_response(&params, n, sample_rate), sample_window(&*params.window, n))
                            };
                            // drops the history like `previous_chunk = None` (filters.rs:187)
                            let status = unsafe {
                                ffi::rr_filter_design(handle.get(), sample_rate, n, response.as_ptr(), window_rel.as_ptr())
                            };
                            if check(status).is_err() {
                                return;
                            }
                        }
                        let mut output_chunk = buf_pool.get_with_capacity(n);
                        let mut n_out = 0usize;
                        let status = unsafe {
                            ffi::rr_filter_enqueue(
                                handle.get(),
                                sample_rate,
                                input_chunk.as_ptr() as *const c_void,
                                n,
                                output_chunk.as_mut_ptr() as *mut c_void,
                                output_chunk.capacity(),
                                &mut n_out,
                            )
                        };
                        if check(status).is_err() {
                            return;
                        }
                        if handle.wait().await.is_err() {
                            return;
                        }
                        drop(input_chunk); // the handle has kept its own copy as `previous_chunk`
                        if n_out == 0 {
                            continue; // first chunk after start / redesign / interrupt (filters.rs:240,260)
                        }
                        unsafe { output_chunk.set_len(n_out) };
                        let Ok(()) = sender
                            .send(Signal::Samples { sample_rate, chunk: output_chunk.finalize() })
                            .await
                        else { return; };
                    }
                    Signal::Event(event) => {
                        if event.is_interrupt() {
                            // `previous_chunk = None` (filters.rs:262-265)
                            if check(unsafe { ffi::rr_filter_reset(handle.get()) }).is_err() {
                                return;
                            }
                        }
                        let Ok(()) = sender.send(Signal::Event(event)).await else { return; };
                    }
                }
            }
        });
        Self { receiver_connector, sender_connector, params: params_send }
    }
    /// Update frequency response and leave window function unchanged
    pub fn update<F>(&self, freq_resp: F)
    where
        F: Fn(isize, f64) -> Complex<f64> + Send + Sync + 'static,
    {
        self.params.send_modify(|params| {
            params.freq_resp = Box::new(freq_resp);
        });
    }
    /// Update frequency response and window function
    pub fn update_with_window<F, W>(&self, freq_resp: F, window: W)
    where
        F: Fn(isize, f64) -> Complex<f64> + Send + Sync + 'static,
        W: Window + Send + Sync + 'static,
    {
        self.params.send_replace(FilterParams { freq_resp: Box::new(freq_resp), window: Box::new(window) });
    }
}
