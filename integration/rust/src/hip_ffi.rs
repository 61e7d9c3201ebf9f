//! `extern "C"` declarations for `include/adsb_hip.h` (libadsb_hip.so, gfx950).
//! NOT COMPILED in the build image (no Rust toolchain there).  It cannot drift from the header
//! unnoticed: tests/test_rust_shim.py parses this file and include/adsb_hip.h and compares every
//! function (name, arity, argument and return types), the status constants and the field order,
//! types, offsets and sizes of the three structs (which tests/abi_host.c also pins with
//! _Static_assert from the C side).
#![allow(dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const ADSB_OK: c_int = 0;
pub const ADSB_ERR_INVALID: c_int = -1;
pub const ADSB_ERR_NO_DEVICE: c_int = -2;
pub const ADSB_ERR_HIP: c_int = -3;
pub const ADSB_ERR_TOO_LONG: c_int = -4;
pub const ADSB_ERR_CAPACITY: c_int = -5;
pub const ADSB_ERR_NOMEM: c_int = -6;
pub const ADSB_ERR_BUSY: c_int = -7;
pub const ADSB_ERR_POISONED: c_int = -8;

/// `adsb_msg`: `ModeSMessage` (src/demod_2400.rs:92-102) + provenance.  40 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct AdsbMsg {
    pub msg: [u8; 14],
    pub len: u8, // 7 | 14 == buffer().len()
    pub try_phase: u8,
    pub score: i32,
    pub j: u32,
    pub chunk: u64,
    pub signal_level: f64,
}

/// `adsb_trial`: one raw trial message, before scoring.  32 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct AdsbTrial {
    pub power: u64,
    pub chunk: u32,
    pub j_tp: u32, // j | try_phase << 24
    pub msg: [u8; 14],
    pub pad: u16,
}

/// `adsb_stats`: counters and timings of the most recent call.  72 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct AdsbStats {
    pub n_samples: u64,
    pub n_chunks: u64,
    pub n_candidates: u64,
    pub n_ap_entries: u64,
    pub n_records: u64,
    pub n_messages: u64,
    pub ms_scan: f32,
    pub ms_match: f32,
    pub ms_records: f32,
    pub ms_total_device: f32,
    pub retries: u32,
    pub ms_scan_exclusive: f32,
}

/// `adsb_multi_stats`: counters and host-clock timings of the capture an `adsb_multi` collected last.  96 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct AdsbMultiStats {
    pub n_samples: u64,
    pub n_chunks: u64,
    pub n_candidates: u64,
    pub n_ap_entries: u64,
    pub n_records: u64,
    pub n_messages: u64,
    pub n_addrs_exchanged: u64,
    pub n_devices: u32,
    pub retries: u32,
    pub ms_wall: f32,
    pub ms_phase1_max: f32,
    pub ms_phase2_max: f32,
    pub ms_phase1_span: f32,
    pub ms_phase2_span: f32,
    pub ms_exchange: f32,
    pub ms_replay: f32,
    pub reserved: f32,
}

#[repr(C)]
pub struct AdsbCtx {
    _private: [u8; 0],
}

/// One capture over several GPUs from one process: one handle, one filter (`adsb_multi_*`).
#[repr(C)]
pub struct AdsbMulti {
    _private: [u8; 0],
}

// (edition 2024, Cargo.toml:7: extern blocks are `unsafe extern`)
#[link(name = "adsb_hip")]
unsafe extern "C" {
    pub fn adsb_create(out: *mut *mut AdsbCtx, device: c_int, max_chunks: usize) -> c_int;
    pub fn adsb_destroy(ctx: *mut AdsbCtx);
    pub fn adsb_set_stream(ctx: *mut AdsbCtx, hip_stream: *mut c_void) -> c_int;
    pub fn adsb_set_profiling(ctx: *mut AdsbCtx, level: c_int) -> c_int;
    pub fn adsb_set_carry_over(ctx: *mut AdsbCtx, enabled: c_int) -> c_int; // opt-in, not the reference's semantics
    pub fn adsb_icao_flush(ctx: *mut AdsbCtx) -> c_int;
    pub fn adsb_to_mag(ctx: *mut AdsbCtx, iq_re_im: *const i16, n: usize, data_out: *mut u16, length_out: *mut usize) -> c_int;
    pub fn adsb_demodulate2400(ctx: *mut AdsbCtx, data: *const u16, length: usize, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_demod_iq(ctx: *mut AdsbCtx, iq_re_im: *const i16, n_samples: usize, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_demod_iq_device(ctx: *mut AdsbCtx, device_iq: *const c_void, n_samples: usize, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_submit_iq_device(ctx: *mut AdsbCtx, device_iq: *const c_void, n_samples: usize) -> c_int;
    pub fn adsb_collect(ctx: *mut AdsbCtx, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_pending(ctx: *const AdsbCtx) -> c_int;
    pub fn adsb_max_in_flight(ctx: *const AdsbCtx) -> c_int;
    pub fn adsb_fetch_messages(ctx: *mut AdsbCtx, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_ring_create(ctx: *mut AdsbCtx, samples_per_slot: usize) -> c_int;
    pub fn adsb_ring_acquire(ctx: *mut AdsbCtx, host_iq_re_im: *mut *mut i16, capacity_samples: *mut usize) -> c_int;
    pub fn adsb_ring_submit(ctx: *mut AdsbCtx, n_samples: usize) -> c_int;
    pub fn adsb_shard_scan(ctx: *mut AdsbCtx, device_iq: *const c_void, n_samples: usize, addrs_out: *mut u32, cap: usize, n_addrs: *mut usize) -> c_int;
    pub fn adsb_shard_finish(ctx: *mut AdsbCtx, extra_addrs: *const u32, n_extra: usize, records_out: *mut AdsbTrial, cap: usize, n_records: *mut usize) -> c_int;
    pub fn adsb_multi_create(out: *mut *mut AdsbMulti, devices: *const c_int, n_devices: c_int, max_chunks_per_device: usize) -> c_int;
    pub fn adsb_multi_destroy(m: *mut AdsbMulti);
    pub fn adsb_multi_device_count(m: *const AdsbMulti) -> c_int;
    pub fn adsb_multi_max_in_flight(m: *const AdsbMulti) -> c_int;
    pub fn adsb_multi_shard_range(n_samples: usize, n_devices: c_int, k: c_int, first_sample: *mut usize, n_samples_k: *mut usize) -> c_int;
    pub fn adsb_multi_icao_flush(m: *mut AdsbMulti) -> c_int;
    pub fn adsb_multi_demod_iq(m: *mut AdsbMulti, iq_re_im: *const i16, n_samples: usize, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_multi_demod_iq_device(m: *mut AdsbMulti, device_iq: *const *const c_void, n_samples: *const usize, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_multi_submit_iq_device(m: *mut AdsbMulti, device_iq: *const *const c_void, n_samples: *const usize) -> c_int;
    pub fn adsb_multi_submit_iq(m: *mut AdsbMulti, iq_re_im: *const i16, n_samples: usize) -> c_int;
    pub fn adsb_multi_host_alloc(m: *mut AdsbMulti, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn adsb_multi_host_free(m: *mut AdsbMulti, host_ptr: *mut c_void) -> c_int;
    pub fn adsb_multi_collect(m: *mut AdsbMulti, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_multi_pending(m: *const AdsbMulti) -> c_int;
    pub fn adsb_multi_fetch_messages(m: *mut AdsbMulti, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_multi_get_stats(m: *const AdsbMulti, out: *mut AdsbMultiStats) -> c_int;
    pub fn adsb_multi_filter_table(m: *const AdsbMulti, out4096: *mut u32) -> c_int;
    pub fn adsb_multi_last_error(m: *const AdsbMulti) -> *const c_char;
    pub fn adsb_multi_set_wait(m: *mut AdsbMulti, mode: c_int) -> c_int;
    pub fn adsb_multi_get_wait(m: *const AdsbMulti) -> c_int;
    pub fn adsb_multi_set_timeout_ms(m: *mut AdsbMulti, ms: u32) -> c_int;
    pub fn adsb_replay_records(filter_table: *mut u32, records: *mut AdsbTrial, n: usize, out: *mut AdsbMsg, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_format_raw(msg: *const AdsbMsg, out: *mut c_char, out_size: usize) -> c_int;
    pub fn adsb_read_test_data(path: *const c_char, iq_re_im: *mut i16, max_samples: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_selftest_mag_digest(ctx: *mut AdsbCtx, first_bits: u32, count: u32, sum_out: *mut u64, xor_out: *mut u64) -> c_int;
    pub fn adsb_selftest_stage_lists(ctx: *mut AdsbCtx, device_iq_re_im: *const c_void, n_samples: usize, cand: *mut u64, cand_cap: usize, n_cand: *mut usize, ap: *mut u64, ap_cap: usize, n_ap: *mut usize) -> c_int;
    pub fn adsb_selftest_gate_stages(ctx: *mut AdsbCtx, device_iq_re_im: *const c_void, n_samples: usize, preamble: *mut u64, preamble_cap: usize, n_preamble: *mut usize, snr: *mut u64, snr_cap: usize, n_snr: *mut usize) -> c_int;
    pub fn adsb_selftest_set_order_polls(ctx: *mut AdsbCtx, polls: u32) -> c_int;
    pub fn adsb_multi_selftest_tune(m: *mut AdsbMulti, fresh_cap: u32, parallel_min: u32, score_mode: u32) -> c_int;
    pub fn adsb_multi_selftest_fail(m: *mut AdsbMulti, captures_from_now: u32, shard: c_int, kind: c_int) -> c_int;
    pub fn adsb_multi_selftest_counters(m: *const AdsbMulti, out8: *mut u64) -> c_int;
    pub fn adsb_selftest_parallel_replay(filter_table: *mut u32, records: *const AdsbTrial, n: usize, runs: c_int, parts: c_int, threads: c_int, out: *mut AdsbMsg, cap: usize, n_out: *mut usize, went_parallel: *mut c_int) -> c_int;
    pub fn adsb_selftest_crc_table(out256: *mut u32) -> c_int;
    pub fn adsb_selftest_learned_union(records: *const AdsbTrial, n: usize, known: *const u32, n_known: usize, out: *mut u32, cap: usize, n_out: *mut usize) -> c_int;
    pub fn adsb_get_stats(ctx: *const AdsbCtx, out: *mut AdsbStats) -> c_int;
    pub fn adsb_host_sorts(ctx: *const AdsbCtx) -> u64;
    pub fn adsb_host_replays(ctx: *const AdsbCtx) -> u64;
    pub fn adsb_host_register(ctx: *mut AdsbCtx, host_ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn adsb_host_unregister(ctx: *mut AdsbCtx, host_ptr: *mut c_void) -> c_int;
    pub fn adsb_host_rematches(ctx: *const AdsbCtx) -> u64;
    pub fn adsb_strerror(status: c_int) -> *const c_char;
    pub fn adsb_last_error(ctx: *const AdsbCtx) -> *const c_char;
    pub fn adsb_version() -> *const c_char;
}
