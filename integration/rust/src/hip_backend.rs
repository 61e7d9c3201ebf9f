//! The three library items of the reference's hot path, re-pointed at libadsb_hip.so.
//! Same names, signatures and failure behaviour as rsadsb/dump1090_rs v0.8.1:
//!   utils::to_mag               src/utils.rs:43   (panics past 131072 samples, like lib.rs:48)
//!   demod_2400::demodulate2400  src/demod_2400.rs:115  (always Ok in the reference)
//!   icao_filter::icao_flush     src/icao_filter.rs:11
//! UNTESTED: written without a Rust toolchain.
use std::sync::Mutex;

use num_complex::Complex;
use once_cell::sync::Lazy;

use crate::demod_2400::{ModeSMessage, MsgLen};
use crate::hip_ffi::*;
use crate::MagnitudeBuffer;

/// One context per process stands in for the ICAO_FILTER_A/B statics (src/icao_filter.rs:8-9).
struct Ctx(*mut AdsbCtx);
unsafe impl Send for Ctx {}

static CTX: Lazy<Mutex<Ctx>> = Lazy::new(|| {
    let mut p: *mut AdsbCtx = std::ptr::null_mut();
    let st = unsafe { adsb_create(&mut p, 0, 1) };
    assert_eq!(st, ADSB_OK, "adsb_create failed: no MI355X / HIP device (libadsb_hip has no CPU fallback)");
    Mutex::new(Ctx(p))
});

pub fn icao_flush() {
    let c = CTX.lock().unwrap();
    unsafe { adsb_icao_flush(c.0) };
}

#[must_use]
pub fn to_mag(data: &[Complex<i16>]) -> MagnitudeBuffer {
    let mut out = MagnitudeBuffer::default();
    let c = CTX.lock().unwrap();
    // Complex<i16> is #[repr(C)] {re, im}: exactly the iq_re_im layout of the ABI
    let st = unsafe { adsb_to_mag(c.0, data.as_ptr().cast::<i16>(), data.len(), out.data.as_mut_ptr(), &mut out.length) };
    assert!(st == ADSB_OK, "to_mag: more than 131072 samples"); // the reference panics here too
    out
}

pub fn demodulate2400(mag: &MagnitudeBuffer) -> Result<Vec<ModeSMessage>, &'static str> {
    let mut raw: Vec<AdsbMsg> = Vec::with_capacity(4096);
    let mut n = 0usize;
    let c = CTX.lock().unwrap();
    let st = unsafe { adsb_demodulate2400(c.0, mag.data.as_ptr(), mag.length, raw.as_mut_ptr(), raw.capacity(), &mut n) };
    if st != ADSB_OK {
        return Err("adsb_demodulate2400 failed");
    }
    unsafe { raw.set_len(n.min(raw.capacity())) };
    Ok(raw
        .iter()
        .map(|m| ModeSMessage {
            msglen: if m.len == 14 { MsgLen::Long } else { MsgLen::Short },
            msg: m.msg,
            signal_level: m.signal_level,
            score: m.score,
        })
        .collect())
}
