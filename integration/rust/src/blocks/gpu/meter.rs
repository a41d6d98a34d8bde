//! [`Meter`]: the front end of `examples/bandwidth_meter/main.rs:53-69` in the example's own order —
//! `FreqShifter` → `Downsampler` → `Filter` → `Overlapper` → `Fourier` — as ONE block whose intermediate
//! streams never leave the device (`rr_meter_*`).
//!
//! Mixer and decimator run as one kernel for any integer or short-period rational ratio (the example's 10 : 1), the
//! `Filter` works on the `Downsampler`'s chunks at the output rate, and one message per overlapped spectrum comes out,
//! ready for `metering::bandwidth`.

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

/// Parameters of the five blocks (same meaning as their constructors' arguments)
pub struct MeterParams {
    /// `FreqShifter::with_precision_and_shift`
    pub precision: f64,
    /// initial frequency shift in hertz
    pub shift: f64,
    /// `Downsampler::with_quality(chunk_len, output_rate, bandwidth, quality)`; `chunk_len` is also the `Filter`'s
    /// chunk length (it receives the `Downsampler`'s chunks)
    pub chunk_len: usize,
    /// output sample rate of the `Downsampler`
    pub output_rate: f64,
    /// aliasing is suppressed below this bandwidth
    pub bandwidth: f64,
    /// `Downsampler` quality (3.0 for `Downsampler::new`)
    pub quality: f64,
    /// `Overlapper::new(overlap)`
    pub overlap: usize,
    /// `Fourier` window: Kaiser β, or `None` for rectangular
    pub fft_kaiser_beta: Option<f64>,
    /// `Fourier::*_center_dc`
    pub center_dc: bool,
}

/// The bandwidth meter's signal path (GPU only)
pub struct Meter<Flt> {
    receiver_connector: ReceiverConnector<Signal<Complex<Flt>>>,
    sender_connector: SenderConnector<Signal<Complex<Flt>>>,
    shift: watch::Sender<f64>,
}

impl_block_trait! { <Flt> Consumer<Signal<Complex<Flt>>> for Meter<Flt> }
impl_block_trait! { <Flt> Producer<Signal<Complex<Flt>>> for Meter<Flt> }

impl<Flt> Meter<Flt>
where
    Flt: GpuFloat,
{
    /// Create the block; `freq_resp` is the `Filter`'s closure (window: `Kaiser::with_null_at_bin(2.0)`)
    pub fn new<F>(params: MeterParams, freq_resp: F) -> Self
    where
        F: Fn(isize, f64) -> Complex<f64> + Send + Sync + 'static,
    {
        let (mut receiver, receiver_connector) = new_receiver::<Signal<Complex<Flt>>>();
        let (sender, sender_connector) = new_sender::<Signal<Complex<Flt>>>();
        let (shift_send, mut shift_recv) = watch::channel(params.shift);
        let c_params = ffi::rr_meter_params {
            dtype: Flt::DTYPE,
            precision: params.precision,
            shift: params.shift,
            output_rate: params.output_rate,
            bandwidth: params.bandwidth,
            quality: params.quality,
            chunk_len: params.chunk_len,
            overlap: params.overlap,
            fft_window: match params.fft_kaiser_beta {
                Some(beta) => ffi::rr_window { kind: ffi::RR_WIN_KAISER, beta },
                None => ffi::rr_window { kind: ffi::RR_WIN_RECTANGULAR, beta: 0.0 },
            },
            center_dc: params.center_dc as c_int,
        };
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::rr_meter_create(&c_params, 0, &mut raw) }).expect("radiorust_amd: no usable MI355X");
        let handle = Handle::new(raw, ffi::rr_meter_destroy);
        // The Filter always sees (output_rate, chunk_len): its closure is sampled once, here (filters.rs:188-199,209-212)
        {
            let n = params.chunk_len;
            let mut response = vec![ffi::rr_c64 { re: 0.0, im: 0.0 }; n];
            let freq_step = params.output_rate / n as f64;
            for i in 0..=(n - 1) / 2 {
                let freq = i as f64 * freq_step;
                let v = freq_resp(i as isize, freq);
                response[i] = ffi::rr_c64 { re: v.re, im: v.im };
                if i > 0 {
                    let v = freq_resp(-(i as isize), -freq);
                    response[n - i] = ffi::rr_c64 { re: v.re, im: v.im };
                }
            }
            let window_rel = sample_window(&Kaiser::with_null_at_bin(2.0), n);
            check(unsafe { ffi::rr_meter_filter_design(handle.get(), response.as_ptr(), window_rel.as_ptr()) })
                .expect("radiorust_amd: no usable MI355X");
        }
        let spectrum_len = params.chunk_len * params.overlap;
        let output_rate = params.output_rate;
        spawn(async move {
            let mut buf_pool = PinnedChunkBufPool::<Complex<Flt>>::new();
            loop {
                let Ok(signal) = receiver.recv().await else { return; };
                match signal {
                    Signal::Samples { sample_rate, chunk: input_chunk } => {
                        if shift_recv.has_changed().unwrap_or(false) {
                            let shift = *shift_recv.borrow_and_update();
                            if check(unsafe { ffi::rr_meter_set_shift(handle.get(), shift) }).is_err() {
                                return;
                            }
                        }
                        let mut frames = 0usize;
                        let status =
                            unsafe { ffi::rr_meter_peek(handle.get(), sample_rate, input_chunk.len(), &mut frames) };
                        if check(status).is_err() {
                            return;
                        }
                        let mut spectra = buf_pool.get_with_capacity((frames * spectrum_len).max(1));
                        let mut n_out = 0usize;
                        // (blocking form: rr_meter has no enqueue entry point; run it off the runtime's threads)
                        let (h, inp, n_in, outp, cap) = (
                            handle.get() as usize,
                            input_chunk.as_ptr() as usize,
                            input_chunk.len(),
                            spectra.as_mut_ptr() as usize,
                            spectra.capacity(),
                        );
                        let done = tokio::task::spawn_blocking(move || {
                            let mut n = 0usize;
                            let status = unsafe {
                                ffi::rr_meter_process(
                                    h as *mut ffi::rr_meter,
                                    sample_rate,
                                    inp as *const c_void,
                                    n_in,
                                    outp as *mut c_void,
                                    cap,
                                    &mut n,
                                )
                            };
                            (status, n)
                        })
                        .await;
                        let Ok((status, n)) = done else { return; };
                        if check(status).is_err() {
                            return;
                        }
                        n_out = n_out.max(n);
                        drop(input_chunk);
                        unsafe { spectra.set_len(n_out) };
                        let mut all = spectra.finalize();
                        while spectrum_len > 0 && all.len() >= spectrum_len {
                            let one = all.separate_beginning(spectrum_len);
                            let Ok(()) = sender.send(Signal::Samples { sample_rate: output_rate, chunk: one }).await
                            else { return; };
                        }
                    }
                    Signal::Event(event) => {
                        // Filter reset on interrupts (filters.rs:262-265); the Overlapper resets on every event and
                        // announces it with SamplesLost (chunks.rs:225-233)
                        let status = unsafe { ffi::rr_meter_event(handle.get(), event.is_interrupt() as c_int) };
                        if check(status).is_err() {
                            return;
                        }
                        let Ok(()) = sender.send(Signal::new_event(crate::blocks::chunks::events::SamplesLost)).await
                        else { return; };
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
