//! MI355X (gfx950) versions of the hot-path blocks, backed by `libradiorust_amd.so`.
//!
//! Each block here has the public API of the CPU block it stands in for — same type name, constructors,
//! methods, `Producer` / `Consumer` implementations — and differs only in the body of its task between
//! `receiver.recv()` and `sender.send()` (the seam marked "no operation here" in [`NopSignal`],
//! `blocks/mod.rs:206-238`): that body calls the C ABI of `include/radiorust_amd.h` instead of doing the
//! arithmetic.  `flow.rs`, `signal.rs` and Tokio stay untouched; Tokio message passing remains the scheduler.
//!
//! | CPU block | GPU block | C handle |
//! |---|---|---|
//! | `blocks::transform::FreqShifter` | [`FreqShifter`] | `rr_freqshifter` |
//! | `blocks::filters::Filter` | [`Filter`] | `rr_filter` |
//! | `blocks::resampling::Downsampler` | [`Downsampler`] | `rr_downsampler` |
//! | `blocks::analysis::Fourier` | [`Fourier`] | `rr_fourier` |
//! | shift → filter → decimate → Fourier in a row | [`Chain`] | `rr_chain` |
//! | `examples/bandwidth_meter`: shift → decimate → filter → overlap → Fourier | [`Meter`] | `rr_meter` |
//!
//! To add this to radiorust: copy this directory to `src/blocks/gpu/`, `build.rs` next to `Cargo.toml`, and
//! add `#[cfg(feature = "mi355x")] pub mod gpu;` to `src/blocks/mod.rs` (feature `mi355x = []`).
//!
//! [`NopSignal`]: crate::blocks::NopSignal

pub mod bufferpool;
pub mod chain;
pub mod downsampler;
pub mod ffi;
pub mod filter;
pub mod fourier;
pub mod freq_shifter;
pub mod meter;

pub use chain::Chain;
pub use downsampler::Downsampler;
pub use filter::Filter;
pub use fourier::Fourier;
pub use freq_shifter::FreqShifter;
pub use meter::Meter;

use crate::numbers::Float;
use crate::windowing::Window;

use std::ffi::CStr;
use std::os::raw::c_int;

/// `Flt` → the backend's dtype code (`numbers.rs:23-42`: the crate is generic over `f32` and `f64`)
pub trait GpuFloat: Float {
    /// `RR_F32` or `RR_F64`
    const DTYPE: c_int;
}
impl GpuFloat for f32 {
    const DTYPE: c_int = ffi::RR_F32;
}
impl GpuFloat for f64 {
    const DTYPE: c_int = ffi::RR_F64;
}

/// The device (or its runtime) failed: the block's task ends, exactly like a task whose peer is gone
/// (`let Ok(..) else { return; }`, `transform.rs:312,355`).
#[derive(Debug)]
pub(crate) struct DeviceLost;

/// Turns a status of the C ABI into the reference's failure modes: what `panic!`s / `assert!`s in the CPU
/// blocks (`resampling.rs:51-56,77-81`) panics here with the library's message; a HIP error ends the task.
pub(crate) fn check(status: c_int) -> Result<(), DeviceLost> {
    match status {
        ffi::RR_OK => Ok(()),
        ffi::RR_ERR_HIP => Err(DeviceLost),
        _ => {
            // RR_ERR_CONTRACT, RR_ERR_CAPACITY, RR_ERR_NEED_DESIGN, RR_ERR_BAD_ARG: a bug on this side of the ABI
            let msg = unsafe { CStr::from_ptr(ffi::rr_last_error_string()) };
            panic!("radiorust_amd: {}", msg.to_string_lossy());
        }
    }
}

/// Owner of one C handle.  A handle holds what the CPU block keeps in its task closure (`transform.rs:307-310`,
/// `filters.rs:161-170`, `resampling.rs:62-67`, `analysis.rs:67-73`): it is used by one task at a time and may
/// move between Tokio's worker threads between calls (every entry point selects the handle's device), hence
/// `Send` but not `Sync`.
pub(crate) struct Handle<T> {
    raw: *mut T,
    destroy: unsafe extern "C" fn(*mut T) -> c_int,
}
unsafe impl<T> Send for Handle<T> {}
impl<T> Handle<T> {
    pub(crate) fn new(raw: *mut T, destroy: unsafe extern "C" fn(*mut T) -> c_int) -> Self {
        assert!(!raw.is_null());
        Self { raw, destroy }
    }
    pub(crate) fn get(&self) -> *mut T {
        self.raw
    }
    /// Waits for everything queued on the handle's stream without blocking a runtime thread — the way
    /// `soapysdr.rs:102-107` treats blocking device calls.
    pub(crate) async fn wait(&self) -> Result<(), DeviceLost> {
        let addr = self.raw as usize;
        let status = tokio::task::spawn_blocking(move || unsafe { ffi::rr_wait(addr as *mut ffi::rr_block) }).await;
        match status {
            Ok(s) => check(s),
            Err(_) => Err(DeviceLost),
        }
    }
}
impl<T> Drop for Handle<T> {
    fn drop(&mut self) {
        unsafe {
            ffi::rr_wait(self.raw as *mut ffi::rr_block);
            (self.destroy)(self.raw);
        }
    }
}

/// `window.relative_value_at(2 (i + 0.5) / n - 1)`, i < n: the positions at which `Filter` and `Fourier`
/// evaluate their window (`filters.rs:209-212`, `analysis.rs:93-94`).  Window trait objects and closures never
/// cross the ABI; the host samples them where the reference evaluates them.
pub(crate) fn sample_window<W: Window + ?Sized>(window: &W, n: usize) -> Vec<f64> {
    let n_flt = n as f64;
    (0..n).map(|i| window.relative_value_at(2.0 * (i as f64 + 0.5) / n_flt - 1.0)).collect()
}
