//! [`Chain`]: `FreqShifter` → `Filter` → `Downsampler` → `Fourier` as ONE block whose intermediate streams
//! never leave the device (`rr_chain_*`).
//!
//! Wiring the four GPU blocks one after the other works, but every hop costs a PCIe round trip.  When they
//! follow each other directly — the front end of `examples/bandwidth_meter/main.rs:51-72` — this block takes the
//! parameters of all four and emits what the last one would emit: one [`Signal::Samples`] per spectrum, at the
//! Downsampler's output rate.  Inside, the library fuses mixer, both filters and the decimation into one
//! kernel and runs the transforms on the decimated stream.
//!
//! The stream entering the Filter is cut into chunks of `filter_len` by the handle (a `Rechunker(filter_len)`,
//! `chunks.rs:42-177`), so input chunks may have any length.

use super::bufferpool::PinnedChunkBufPool;
use super::{check, ffi, sample_window, GpuFloat, Handle};
use crate::flow::*;
use crate::impl_block_trait;
use crate::numbers::*;
use crate::signal::*;
use crate::windowing::{Kaiser, Window};

use tokio::sync::watch;
use tokio::task::spawn;

use std::os::raw::{c_int, c_void};
use std::ptr;

trait FreqRespFunc: Fn(isize, f64) -> Complex<f64> {}
impl<T: ?Sized> FreqRespFunc for T where T: Fn(isize, f64) -> Complex<f64> {}

/// Parameters of the four blocks (same meaning as their constructors' arguments)
pub struct ChainParams {
    /// `FreqShifter::with_precision_and_shift`
    pub precision: f64,
    /// initial frequency shift in hertz
    pub shift: f64,
    /// chunk length (= tap count) the `Filter` works with
    pub filter_len: usize,
    /// `Downsampler::with_quality`
    pub output_rate: f64,
    /// `Downsampler`: aliasing is suppressed below this bandwidth
    pub bandwidth: f64,
    /// `Downsampler` quality (3.0 for `Downsampler::new`)
    pub quality: f64,
    /// `Downsampler` output chunk length = `Fourier` length
    pub fft_len: usize,
    /// `Fourier` window: Kaiser β, or `None` for rectangular
    pub fft_kaiser_beta: Option<f64>,
    /// `Fourier::*_center_dc`
    pub center_dc: bool,
}

/// The fused front end (GPU only)
pub struct Chain<Flt> {
    receiver_connector: ReceiverConnector<Signal<Complex<Flt>>>,
    sender_connector: SenderConnector<Signal<Complex<Flt>>>,
    shift: watch::Sender<f64>,
}

impl_block_trait! { <Flt> Consumer<Signal<Complex<Flt>>> for Chain<Flt> }
impl_block_trait! { <Flt> Producer<Signal<Complex<Flt>>> for Chain<Flt> }

impl<Flt> Chain<Flt>
where
    Flt: GpuFloat,
{
    /// Create the block; `freq_resp` is the `Filter`'s closure (its window is `Kaiser::with_null_at_bin(2.0)`)
    pub fn new<F>(params: ChainParams, freq_resp: F) -> Self
    where
        F: Fn(isize, f64) -> Complex<f64> + Send + Sync + 'static,
    {
        Self::with_filter_window(params, freq_resp, Kaiser::with_null_at_bin(2.0))
    }
    /// Create the block with an explicit `Filter` window
    pub fn with_filter_window<F, W>(params: ChainParams, freq_resp: F, filter_window: W) -> Self
    where
        F: Fn(isize, f64) -> Complex<f64> + Send + Sync + 'static,
        W: Window + Send + Sync + 'static,
    {
        let (mut receiver, receiver_connector) = new_receiver::<Signal<Complex<Flt>>>();
        let (sender, sender_connector) = new_sender::<Signal<Complex<Flt>>>();
        let (shift_send, mut shift_recv) = watch::channel(params.shift);
        let c_params = ffi::rr_chain_params {
            dtype: Flt::DTYPE,
            precision: params.precision,
            shift: params.shift,
            filter_len: params.filter_len,
            output_rate: params.output_rate,
            bandwidth: params.bandwidth,
            quality: params.quality,
            fft_len: params.fft_len,
            fft_window: match params.fft_kaiser_beta {
                Some(beta) => ffi::rr_window { kind: ffi::RR_WIN_KAISER, beta },
                None => ffi::rr_window { kind: ffi::RR_WIN_RECTANGULAR, beta: 0.0 },
            },
            center_dc: params.center_dc as c_int,
            allow_fused: 1,
        };
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::rr_chain_create(&c_params, 0, &mut raw) }).expect("radiorust_amd: no usable MI355X");
        let handle = Handle::new(raw, ffi::rr_chain_destroy);
        let filter_len = params.filter_len;
        let fft_len = params.fft_len;
        let output_rate = params.output_rate;
        let freq_resp: Box<dyn FreqRespFunc + Send + Sync> = Box::new(freq_resp);
        spawn(async move {
            let mut buf_pool = PinnedChunkBufPool::<Complex<Flt>>::new();
            let mut prev_sample_rate: Option<f64> = None;
            loop {
                let Ok(signal) = receiver.recv().await else { return; };
                match signal {
                    Signal::Samples { sample_rate, chunk: input_chunk } => {
                        // The Rechunker inside the chain holds < filter_len samples between calls.  A chunk at another
                        // sample rate makes it drop them and report SamplesLost - an interrupt for the Filter behind it
                        // (chunks.rs:72-92; the header's protocol around rr_chain_pending)
                        if prev_sample_rate.map_or(false, |r| r != sample_rate) {
                            let mut held = 0usize;
                            if check(unsafe { ffi::rr_chain_pending(handle.get(), &mut held) }).is_err() {
                                return;
                            }
                            if held > 0 {
                                if check(unsafe { ffi::rr_chain_interrupt(handle.get()) }).is_err() {
                                    return;
                                }
                                let Ok(()) = sender.send(Signal::new_event(crate::blocks::chunks::events::SamplesLost)).await
                                else { return; };
                            }
                        }
                        prev_sample_rate = Some(sample_rate);
                        if shift_recv.has_changed().unwrap_or(false) {
                            let shift = *shift_recv.borrow_and_update();
                            if check(unsafe { ffi::rr_chain_set_shift(handle.get(), shift) }).is_err() {
                                return;
                            }
                        }
                        // Filter design for this sample rate (filters.rs:184-225), closure evaluated here
                        let mut needed: c_int = 0;
                        if check(unsafe { ffi::rr_chain_filter_needs_design(handle.get(), sample_rate, &mut needed) }).is_err() {
                            return;
                        }
                        if needed != 0 {
                            let n = filter_len;
                            let mut response = vec![ffi::rr_c64 { re: 0.0, im: 0.0 }; n];
                            let freq_step = sample_rate / n as f64;
                            for i in 0..=(n - 1) / 2 {
                                let freq = i as f64 * freq_step;
                                let v = freq_resp(i as isize, freq);
                                response[i] = ffi::rr_c64 { re: v.re, im: v.im };
                                if i > 0 {
                                    let v = freq_resp(-(i as isize), -freq);
                                    response[n - i] = ffi::rr_c64 { re: v.re, im: v.im };
                                }
                            }
                            let window_rel = sample_window(&filter_window, n);
                            let status = unsafe {
                                ffi::rr_chain_filter_design(handle.get(), sample_rate, response.as_ptr(), window_rel.as_ptr())
                            };
                            if check(status).is_err() {
                                return;
                            }
                        }
                        let mut frames = 0usize;
                        let status =
                            unsafe { ffi::rr_chain_peek(handle.get(), sample_rate, input_chunk.len(), &mut frames) };
                        if check(status).is_err() {
                            return;
                        }
                        let mut spectra = buf_pool.get_with_capacity((frames * fft_len).max(1));
                        let mut n_out = 0usize;
                        let status = unsafe {
                            ffi::rr_chain_enqueue(
                                handle.get(),
                                sample_rate,
                                input_chunk.as_ptr() as *const c_void,
                                input_chunk.len(),
                                spectra.as_mut_ptr() as *mut c_void,
                                spectra.capacity(),
                                &mut n_out,
                            )
                        };
                        if check(status).is_err() {
                            return;
                        }
                        if handle.wait().await.is_err() {
                            return;
                        }
                        drop(input_chunk);
                        unsafe { spectra.set_len(n_out) };
                        // one message per spectrum, as `Fourier` would send them (zero-copy views of one buffer)
                        let mut all = spectra.finalize();
                        while all.len() >= fft_len && fft_len > 0 {
                            let one = all.separate_beginning(fft_len);
                            let Ok(()) = sender.send(Signal::Samples { sample_rate: output_rate, chunk: one }).await
                            else { return; };
                        }
                    }
                    Signal::Event(event) => {
                        // ANY event while the Rechunker holds samples: they are dropped and SamplesLost goes out in front of
                        // the event (chunks.rs:80-88) - which interrupts the Filter; otherwise only an interrupting event does
                        let mut held = 0usize;
                        if check(unsafe { ffi::rr_chain_pending(handle.get(), &mut held) }).is_err() {
                            return;
                        }
                        if held > 0 {
                            if check(unsafe { ffi::rr_chain_interrupt(handle.get()) }).is_err() {
                                return;
                            }
                            let Ok(()) = sender.send(Signal::new_event(crate::blocks::chunks::events::SamplesLost)).await
                            else { return; };
                        } else if event.is_interrupt() {
                            // the Filter drops its history, the other three keep their state
                            // (filters.rs:262-265; transform.rs:357-359, resampling.rs:135-137, analysis.rs:122-124)
                            if check(unsafe { ffi::rr_chain_interrupt(handle.get()) }).is_err() {
                                return;
                            }
                        }
                        let Ok(()) = sender.send(Signal::Event(event)).await else { return; };
                    }
                }
            }
        });
        Self { receiver_connector, sender_connector, shift: shift_send }
    }
    /// Get current frequency shift
    pub fn shift(&self) -> f64 {
        *self.shift.borrow()
    }
    /// Set frequency shift (phase-continuous, effective from the next chunk)
    pub fn set_shift(&self, shift: f64) {
        self.shift.send_replace(shift);
    }
}
