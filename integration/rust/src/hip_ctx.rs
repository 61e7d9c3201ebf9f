//! The process-wide `adsb_ctx` that stands in for the reference's `ICAO_FILTER_A/B` statics
//! (src/icao_filter.rs:8-9): one context = one stream of IQ with its own ICAO filter.
//! NOT COMPILED in the build image (no Rust toolchain there); std only, MSRV 1.88 (Cargo.toml:9).
use std::sync::{LazyLock, Mutex};

use crate::hip_ffi::{adsb_create, AdsbCtx, ADSB_OK};

struct Ctx(*mut AdsbCtx);
// the context is only ever used under the mutex
unsafe impl Send for Ctx {}

static CTX: LazyLock<Mutex<Ctx>> = LazyLock::new(|| {
    let mut p: *mut AdsbCtx = std::ptr::null_mut();
    // device 0 (ADSB_HIP_DEVICE overrides), lists sized for one 131072-sample buffer per call
    let device = std::env::var("ADSB_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
    let st = unsafe { adsb_create(&mut p, device, 1) };
    assert_eq!(st, ADSB_OK, "adsb_create failed: no MI355X / HIP device (libadsb_hip has no CPU fallback)");
    Mutex::new(Ctx(p))
});

/// Run `f` with the process context (serialised: a context is not thread-safe, and the reference's
/// statics are behind mutexes too).
pub(crate) fn with_ctx<R>(f: impl FnOnce(*mut AdsbCtx) -> R) -> R {
    let guard = CTX.lock().unwrap_or_else(|e| e.into_inner());
    f(guard.0)
}
