//! [`ChunkBufPool`] re-backed by pinned (page-locked) host memory (`bufferpool.rs:187-222`).
//!
//! A [`ChunkBuf`] derefs to `Vec<T>` and a finalized [`Chunk`] recycles its `Vec` to the pool it came from
//! (`bufferpool.rs:44-48,213-222`), so the storage stays an ordinary `Vec<T>`; what this pool adds is that each
//! allocation is registered with the HIP runtime once (`rr_host_register`), which makes the copies of
//! `rr_*_enqueue` true asynchronous DMA transfers on the handle's stream.  Buffers come back through the
//! reference's own recycling channel, so after the first few messages every buffer handed out is already
//! registered.

use super::ffi;
use crate::bufferpool::{ChunkBuf, ChunkBufPool};

use std::collections::HashMap;
use std::mem::size_of;
use std::os::raw::c_void;

/// Pool of [`ChunkBuf`]s whose storage is page-locked
pub struct PinnedChunkBufPool<T> {
    inner: ChunkBufPool<T>,
    /// address of a registered allocation → its size in bytes
    registered: HashMap<usize, usize>,
}

impl<T> PinnedChunkBufPool<T> {
    /// Create a new pool
    pub fn new() -> Self {
        Self { inner: ChunkBufPool::new(), registered: HashMap::new() }
    }
    /// Get an empty [`ChunkBuf`] with at least `capacity` elements of page-locked storage
    pub fn get_with_capacity(&mut self, capacity: usize) -> ChunkBuf<T> {
        let mut buf = self.inner.get_with_capacity(capacity);
        if buf.capacity() < capacity {
            // a recycled buffer that is too small: growing it moves it, the old registration goes
            self.forget(buf.as_ptr() as usize);
            buf.reserve(capacity);
        }
        let addr = buf.as_ptr() as usize;
        let bytes = buf.capacity() * size_of::<T>();
        if bytes != 0 && self.registered.get(&addr) != Some(&bytes) {
            self.forget(addr);
            // failure to pin is not an error: the copy is then staged by the runtime (slower, still correct)
            if unsafe { ffi::rr_host_register(addr as *mut c_void, bytes) } == ffi::RR_OK {
                self.registered.insert(addr, bytes);
            }
        }
        buf
    }
    fn forget(&mut self, addr: usize) {
        if self.registered.remove(&addr).is_some() {
            unsafe { ffi::rr_host_unregister(addr as *mut c_void) };
        }
    }
}

impl<T> Default for PinnedChunkBufPool<T> {
    fn default() -> Self {
        Self::new()
    }
}

impl<T> Drop for PinnedChunkBufPool<T> {
    fn drop(&mut self) {
        // Chunks still in flight keep their Vec alive; unregistering only ends the page lock
        for (&addr, _) in self.registered.iter() {
            unsafe { ffi::rr_host_unregister(addr as *mut c_void) };
        }
    }
}
