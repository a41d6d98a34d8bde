//! [`FreqShifter`] on the MI355X: drop-in for `blocks::transform::FreqShifter` (`transform.rs:266-391`).

use super::bufferpool::PinnedChunkBufPool;
use super::{check, ffi, GpuFloat, Handle};
use crate::flow::*;
use crate::impl_block_trait;
use crate::numbers::*;
use crate::signal::*;

use tokio::sync::watch;
use tokio::task::spawn;

use std::os::raw::c_void;
use std::ptr;

/// Complex oscillator and mixer, which shifts all frequencies in an I/Q stream (GPU version)
///
/// The phase table, its index and the phase-continuous retune live in the device handle
/// (`rr_freqshifter`: `transform.rs:307-340`); the table is built on the host in `Flt` exactly as the CPU block
/// builds it.
pub struct FreqShifter<Flt> {
    receiver_connector: ReceiverConnector<Signal<Complex<Flt>>>,
    sender_connector: SenderConnector<Signal<Complex<Flt>>>,
    precision: f64,
    shift: watch::Sender<f64>,
}

impl_block_trait! { <Flt> Consumer<Signal<Complex<Flt>>> for FreqShifter<Flt> }
impl_block_trait! { <Flt> Producer<Signal<Complex<Flt>>> for FreqShifter<Flt> }

impl<Flt> FreqShifter<Flt>
where
    Flt: GpuFloat,
{
    /// Create new `FreqShifter` block with 1 Hz precision and initial frequency shift of zero
    pub fn new() -> Self {
        Self::with_precision_and_shift(1.0, 0.0)
    }
    /// Create new `FreqShifter` block with 1 Hz precision and given initial frequency `shift` in hertz
    pub fn with_shift(shift: f64) -> Self {
        Self::with_precision_and_shift(1.0, shift)
    }
    /// Create new `FreqShifter` block with given `precision` in hertz and an initial shift of zero
    pub fn with_precision(precision: f64) -> Self {
        Self::with_precision_and_shift(precision, 0.0)
    }
    /// Create new `FreqShifter` block with given `precision` and `shift`, both in hertz
    pub fn with_precision_and_shift(precision: f64, shift: f64) -> Self {
        let (mut receiver, receiver_connector) = new_receiver::<Signal<Complex<Flt>>>();
        let (sender, sender_connector) = new_sender::<Signal<Complex<Flt>>>();
        let (shift_send, mut shift_recv) = watch::channel(shift);
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::rr_freqshifter_create(Flt::DTYPE, precision, shift, 0, &mut raw) })
            .expect("radiorust_amd: no usable MI355X");
        let handle = Handle::new(raw, ffi::rr_freqshifter_destroy);
        spawn(async move {
            let mut buf_pool = PinnedChunkBufPool::<Complex<Flt>>::new();
            loop {
                let Ok(signal) = receiver.recv().await else { return; };
                match signal {
                    Signal::Samples { sample_rate, chunk: input_chunk } => {
                        // `watch` semantics: a new shift applies from this message on (transform.rs:318)
                        if shift_recv.has_changed().unwrap_or(false) {
                            let shift = *shift_recv.borrow_and_update();
                            if check(unsafe { ffi::rr_freqshifter_set_shift(handle.get(), shift) }).is_err() {
                                return;
                            }
                        }
                        let n = input_chunk.len();
                        let mut output_chunk = buf_pool.get_with_capacity(n);
                        let mut n_out = 0usize;
                        // queue H2D copy, kernel and D2H copy on the handle's stream ..
                        let status = unsafe {
                            ffi::rr_freqshifter_enqueue(
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
                        // .. and wait for them off the runtime's threads; the input chunk is borrowed until then
                        if handle.wait().await.is_err() {
                            return;
                        }
                        drop(input_chunk);
                        unsafe { output_chunk.set_len(n_out) };
                        let Ok(()) = sender
                            .send(Signal::Samples { sample_rate, chunk: output_chunk.finalize() })
                            .await
                        else { return; };
                    }
                    event @ Signal::Event { .. } => {
                        let Ok(()) = sender.send(event).await else { return; };
                    }
                }
            }
        });
        Self { receiver_connector, sender_connector, precision, shift: shift_send }
    }
    /// Get frequency precision in hertz
    pub fn precision(&self) -> f64 {
        self.precision
    }
    /// Get current frequency shift
    pub fn shift(&self) -> f64 {
        *self.shift.borrow()
    }
    /// Set frequency shift
    pub fn set_shift(&self, shift: f64) {
        self.shift.send_replace(shift);
    }
    /// Update frequency shift
    pub fn update_shift<F: FnOnce(&mut f64)>(&self, modify: F) {
        self.shift.send_modify(modify);
    }
}
