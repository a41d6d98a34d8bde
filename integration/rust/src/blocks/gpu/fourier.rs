//! [`Fourier`] on the MI355X: drop-in for `blocks::analysis::Fourier` (`analysis.rs:26-133`).

use super::bufferpool::PinnedChunkBufPool;
use super::{check, ffi, sample_window, GpuFloat, Handle};
use crate::flow::*;
use crate::impl_block_trait;
use crate::numbers::*;
use crate::signal::*;
use crate::windowing::{self, Window};

use tokio::task::spawn;

use std::os::raw::c_void;
use std::ptr;

/// Block performing a Fourier analysis (GPU version)
///
/// Every received chunk is windowed (window scaled to unit mean energy, `analysis.rs:88-101`) and transformed
/// (forward, unnormalised); with `center_dc` the DC bin is rotated to index `n / 2`.  Any chunk length works.
/// The window is an arbitrary trait object: it is sampled on the host whenever the chunk length changes
/// (`analysis.rs:82-104`) and handed over as an array (`rr_fourier_set_sampled_window`).
pub struct Fourier<Flt> {
    receiver_connector: ReceiverConnector<Signal<Complex<Flt>>>,
    sender_connector: SenderConnector<Signal<Complex<Flt>>>,
}

impl_block_trait! { <Flt> Consumer<Signal<Complex<Flt>>> for Fourier<Flt> }
impl_block_trait! { <Flt> Producer<Signal<Complex<Flt>>> for Fourier<Flt> }

impl<Flt> Fourier<Flt>
where
    Flt: GpuFloat,
{
    /// Create `Fourier` block without windowing
    pub fn new() -> Self {
        Self::new_internal(windowing::Rectangular, false)
    }
    /// Create `Fourier` block without windowing but rotating DC to center
    pub fn new_center_dc() -> Self {
        Self::new_internal(windowing::Rectangular, true)
    }
    /// Create `Fourier` block with windowing
    pub fn with_window<W>(window: W) -> Self
    where
        W: Window + Send + 'static,
    {
        Self::new_internal(window, false)
    }
    /// Create `Fourier` block with windowing and rotating DC to center
    pub fn with_window_center_dc<W>(window: W) -> Self
    where
        W: Window + Send + 'static,
    {
        Self::new_internal(window, true)
    }
    fn new_internal<W>(window: W, center_dc: bool) -> Self
    where
        W: Window + Send + 'static,
    {
        let (mut receiver, receiver_connector) = new_receiver::<Signal<Complex<Flt>>>();
        let (sender, sender_connector) = new_sender::<Signal<Complex<Flt>>>();
        let sampled = ffi::rr_window { kind: ffi::RR_WIN_SAMPLED, beta: 0.0 };
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::rr_fourier_create(Flt::DTYPE, &sampled, center_dc as _, 0, &mut raw) })
            .expect("radiorust_amd: no usable MI355X");
        let handle = Handle::new(raw, ffi::rr_fourier_destroy);
        spawn(async move {
            let mut buf_pool = PinnedChunkBufPool::<Complex<Flt>>::new();
            let mut previous_chunk_len: Option<usize> = None;
            loop {
                let Ok(signal) = receiver.recv().await else { return; };
                match signal {
                    Signal::Samples { sample_rate, chunk: input_chunk } => {
                        let n = input_chunk.len();
                        if n == 0 {
                            continue;
                        }
                        if Some(n) != previous_chunk_len {
                            let window_rel = sample_window(&window, n);
                            let status =
                                unsafe { ffi::rr_fourier_set_sampled_window(handle.get(), n, window_rel.as_ptr()) };
                            if check(status).is_err() {
                                return;
                            }
                            previous_chunk_len = Some(n);
                        }
                        let mut output_chunk = buf_pool.get_with_capacity(n);
                        let mut n_out = 0usize;
                        let status = unsafe {
                            ffi::rr_fourier_enqueue(
                                handle.get(),
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
        Self { receiver_connector, sender_connector }
    }
}
