//! [`Downsampler`] on the MI355X: drop-in for `blocks::resampling::Downsampler` (`resampling.rs:14-146`).

use super::bufferpool::PinnedChunkBufPool;
use super::{check, ffi, GpuFloat, Handle};
use crate::bufferpool::ChunkBufPool;
use crate::flow::*;
use crate::impl_block_trait;
use crate::numbers::*;
use crate::signal::*;

use tokio::task::spawn;

use std::os::raw::c_void;
use std::ptr;

/// Reduce sample rate (GPU version)
///
/// The impulse response (`resampling.rs:75-102`), the ring buffer and the `pos` schedule (`:103-112`) live in
/// the device handle, which produces the raw decimated stream; this task regroups it into chunks of
/// `output_chunk_len` exactly like `resampling.rs:121-131`: a chunk is sent as soon as it is full, a partly
/// filled one waits for the next message.
pub struct Downsampler<Flt> {
    receiver_connector: ReceiverConnector<Signal<Complex<Flt>>>,
    sender_connector: SenderConnector<Signal<Complex<Flt>>>,
}

impl_block_trait! { <Flt> Consumer<Signal<Complex<Flt>>> for Downsampler<Flt> }
impl_block_trait! { <Flt> Producer<Signal<Complex<Flt>>> for Downsampler<Flt> }

impl<Flt> Downsampler<Flt>
where
    Flt: GpuFloat,
{
    /// Create new `Downsampler` block (`quality` = 3.0)
    ///
    /// Connected [`Producer`]s must emit [`Signal::Samples`] with a sample rate equal to or higher than
    /// `output_rate`; otherwise a panic occurs.  Aliasing is suppressed for frequencies lower than `bandwidth`.
    pub fn new(output_chunk_len: usize, output_rate: f64, bandwidth: f64) -> Self {
        Self::with_quality(output_chunk_len, output_rate, bandwidth, 3.0)
    }
    /// Create new `Downsampler` block with `quality` setting (equal to or greater than `1.0`)
    pub fn with_quality(output_chunk_len: usize, output_rate: f64, bandwidth: f64, quality: f64) -> Self {
        assert!(output_rate >= 0.0, "output sample rate must be positive");
        assert!(bandwidth >= 0.0, "bandwidth must be positive");
        assert!(bandwidth < output_rate, "bandwidth must be smaller than output sample rate");
        let (mut receiver, receiver_connector) = new_receiver::<Signal<Complex<Flt>>>();
        let (sender, sender_connector) = new_sender::<Signal<Complex<Flt>>>();
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::rr_downsampler_create(Flt::DTYPE, output_rate, bandwidth, quality, 0, &mut raw) })
            .expect("radiorust_amd: no usable MI355X");
        let handle = Handle::new(raw, ffi::rr_downsampler_destroy);
        // the chunks that go out (ordinary pool, filled by memcpy from the staging buffer) ..
        let mut buf_pool = ChunkBufPool::<Complex<Flt>>::new();
        let mut output_chunk = buf_pool.get_with_capacity(output_chunk_len);
        spawn(async move {
            // .. and the pinned buffers the device writes one message's raw outputs into
            let mut stage_pool = PinnedChunkBufPool::<Complex<Flt>>::new();
            loop {
                let Ok(signal) = receiver.recv().await else { return; };
                match signal {
                    Signal::Samples { sample_rate: input_rate, chunk: input_chunk } => {
                        // the rate contract of resampling.rs:77-81 comes back as RR_ERR_CONTRACT -> panic
                        let mut produce = 0usize;
                        let status = unsafe {
                            ffi::rr_downsampler_peek(handle.get(), input_rate, input_chunk.len(), &mut produce)
                        };
                        if check(status).is_err() {
                            return;
                        }
                        let mut stage = stage_pool.get_with_capacity(produce.max(1));
                        let mut n_out = 0usize;
                        let status = unsafe {
                            ffi::rr_downsampler_enqueue(
                                handle.get(),
                                input_rate,
                                input_chunk.as_ptr() as *const c_void,
                                input_chunk.len(),
                                stage.as_mut_ptr() as *mut c_void,
                                stage.capacity(),
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
                        unsafe { stage.set_len(n_out) };
                        // regroup (resampling.rs:121-131)
                        let mut rest: &[Complex<Flt>] = &stage;
                        while !rest.is_empty() {
                            let take = (output_chunk_len - output_chunk.len()).min(rest.len());
                            output_chunk.extend_from_slice(&rest[..take]);
                            rest = &rest[take..];
                            if output_chunk.len() >= output_chunk_len {
                                let Ok(()) = sender
                                    .send(Signal::Samples { sample_rate: output_rate, chunk: output_chunk.finalize() })
                                    .await
                                else { return; };
                                output_chunk = buf_pool.get_with_capacity(output_chunk_len);
                            }
                        }
                    }
                    event @ Signal::Event { .. } => {
                        // no reset, the partly filled chunk stays (resampling.rs:135-137)
                        let Ok(()) = sender.send(event).await else { return; };
                    }
                }
            }
        });
        Self { receiver_connector, sender_connector }
    }
}
