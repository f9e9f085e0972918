//! Raw bindings to `libeigentrust_hip.so`: one declaration per function of
//! `include/eigentrust_hip.h`, same order.  The reference has no FFI on this path; its
//! `EigenTrustSet` (eigentrust-zk/src/circuits/dynamic_sets/native.rs:109-392) lives in a crate
//! that forbids unsafe code (eigentrust-zk/src/lib.rs:9), so the binding is this separate crate.
//!
//! NOT COMPILED in the image this repository is built in (it has no Rust toolchain); a test keeps
//! the declarations below in step with the header by name and argument count.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

/// Canonical 32-byte little-endian integer below r, i.e. `Fr::to_repr()` (utils.rs:327-329).
#[repr(C)]
#[derive(Clone, Copy, PartialEq, Eq, Debug, Default)]
pub struct et_fr {
    pub b: [u8; 32],
}

pub const ET_OK: c_int = 0;
pub const ET_E_DUP_MEMBER: c_int = -1;
pub const ET_E_SET_FULL: c_int = -2;
pub const ET_E_NOT_MEMBER: c_int = -3;
pub const ET_E_INSUFFICIENT_PEERS: c_int = -4;
pub const ET_E_SUM_MISMATCH: c_int = -5;
pub const ET_E_NONCANONICAL_FR: c_int = -6;
pub const ET_E_INVALID_ARG: c_int = -7;
pub const ET_E_HIP: c_int = -8;
pub const ET_E_NO_DEVICE: c_int = -9;
pub const ET_E_BUFFER_TOO_SMALL: c_int = -10;
pub const ET_E_OUT_OF_RANGE: c_int = -11;
pub const ET_E_UNSUPPORTED: c_int = -12;
pub const ET_E_THRESHOLD_RANGE: c_int = -13;
pub const ET_E_RATIO_DIGITS: c_int = -14;
pub const ET_E_RATIO_MISMATCH: c_int = -15;
pub const ET_E_ZERO_LIMB: c_int = -16;
pub const ET_E_BAD_DOMAIN: c_int = -17;
pub const ET_E_RATIONAL_TOO_LARGE: c_int = -18;
pub const ET_E_PARSE: c_int = -19;
pub const ET_E_BAD_SIGNATURE: c_int = -20;
pub const ET_E_BAD_ABOUT: c_int = -21;
pub const ET_E_RCCL: c_int = -22;

pub const ET_VARIANT_FR: u32 = 1;
pub const ET_VARIANT_F64: u32 = 2;
pub const ET_INIT_PLAIN: u32 = 1;
pub const ET_STEP_FINAL: u32 = 1;
pub const ET_STEP_NO_STATS: u32 = 2;

#[repr(C)]
pub struct et_set {
    _private: [u8; 0],
}
#[repr(C)]
pub struct et_graph {
    _private: [u8; 0],
}
#[repr(C)]
pub struct et_plan {
    _private: [u8; 0],
}
#[repr(C)]
pub struct et_batch {
    _private: [u8; 0],
}
#[repr(C)]
pub struct et_sharded {
    _private: [u8; 0],
}
#[repr(C)]
pub struct et_dplan {
    _private: [u8; 0],
}
pub const ET_EXCHANGE_AUTO: u32 = 0;
pub const ET_EXCHANGE_RCCL: u32 = 1;
pub const ET_EXCHANGE_COPY: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct et_build_opts {
    pub n_seg: u32,
    pub seg: u32,
    pub long_row: u32,
    pub target_work: u32,
    pub variants: u32,
    pub device: i32,
    pub n_chunks: u32,
    pub flags: u32,
    pub xcd_cost: [u32; 8],
}
pub const ET_BUILD_FULL_VECTOR: u32 = 1;
pub const ET_BUILD_REUSED: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct et_graph_info {
    pub n: u32,
    pub n_valid: u32,
    pub n_dangling: u32,
    pub n_seg: u32,
    pub seg: u32,
    pub nnz: u64,
    pub nnz_local: u64,
    pub padded_local: u64,
    pub n_slices: u32,
    pub n_work: u32,
    pub rows_cap: u32,
    pub nd_cap: u32,
    pub vec_len: u64,
    pub device_bytes: u64,
    pub f64_sweep_blocks: u32,
    pub exact_sweep_blocks: u32,
    pub n_chunks: u32,
    pub chunk_rows: u32,
    pub gather: u32,
    pub full_len: u64,
    pub xcd_cost: [u32; 8],
}

/// What a share sends to and receives from the others between two steps (entries, not bytes).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct et_exchange_info {
    pub gather: u32,
    pub n_seg: u32,
    pub seg: u32,
    pub own_off: u32,
    pub own_len: u32,
    pub vec_len: u64,
    pub full_len: u64,
    pub recv_off: [u32; 8],
    pub recv_count: [u32; 8],
    pub send_off: [u32; 8],
    pub send_count: [u32; 8],
    pub send_total: u64,
    pub self_off: u32,
    pub self_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct et_slice_desc {
    pub off: u32,
    pub len: u32,
    pub row0: u32,
    pub nrows: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct et_work_desc {
    pub slice_begin: u32,
    pub slice_end: u32,
    pub flags: u32,
    pub dslot: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct et_plan_arrays {
    pub pos_of_slot: *const u32,
    pub slices: *const et_slice_desc,
    pub work: *const et_work_desc,
    pub col: *const u32,
    pub value_u32: *const u32,
    pub value_fr: *const et_fr,
    pub scale_fr: *const et_fr,
    pub scale_f64: *const f64,
    pub inv_nm1_w: et_fr,
    pub n_active: u32,
    pub n_active_pad: u32,
    pub n_dangling_pad: u32,
    pub n_dangling_local: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct et_sweep_stats {
    pub blocks: u32,
    pub clustered: u32,
    pub edges: u64,
    pub edges_most: u64,
    pub line_fills: u64,
    pub line_fills_most: u64,
    pub chunk_span_most: u64,
    pub xcd_line_fills: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct et_iter_stats {
    pub sum: et_fr,
    pub changed: u64,
    pub residual_l1: f64,
    pub sum_f64: f64,
}

/// `Score` (eigentrust/src/lib.rs:207-233): magnitudes big-endian, right-aligned in 32 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct et_score {
    pub address: [u8; 20],
    pub score_fr: [u8; 32],
    pub numerator: [u8; 32],
    pub denominator: [u8; 32],
    pub score: [u8; 32],
}

/// `SignedAttestationRaw` (eigentrust/src/attestation.rs:283-330).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct et_attestation {
    pub about: [u8; 20],
    pub domain: [u8; 20],
    pub value: u8,
    pub message: [u8; 32],
    pub sig_r: [u8; 32],
    pub sig_s: [u8; 32],
    pub rec_id: u8,
}

extern "C" {
    pub fn et_strerror(code: c_int) -> *const c_char;
    pub fn et_last_error_message() -> *const c_char;
    pub fn et_abi_version() -> u32;
    pub fn et_device_count() -> c_int;

    // ---- set API: EigenTrustSet (native.rs:165-392) ----
    pub fn et_set_new(num_neighbours: u32, num_iterations: u32, initial_score_lo: u64, initial_score_hi: u64,
                      domain: *const et_fr, out: *mut *mut et_set) -> c_int;
    pub fn et_set_free(set: *mut et_set);
    pub fn et_set_add_member(set: *mut et_set, addr: *const et_fr) -> c_int;
    pub fn et_set_remove_member(set: *mut et_set, addr: *const et_fr) -> c_int;
    pub fn et_set_update_op(set: *mut et_set, from: *const et_fr, scores: *const et_fr) -> c_int;
    pub fn et_set_update_op_sparse(set: *mut et_set, from: *const et_fr, count: u32, about_slot: *const u32,
                                   values: *const et_fr) -> c_int;
    pub fn et_set_update_op_u32(set: *mut et_set, from_slot: u32, count: u32, about_slot: *const u32,
                                values: *const u32) -> c_int;
    pub fn et_set_members(set: *const et_set, addrs: *mut et_fr, scores: *mut et_fr) -> c_int;
    pub fn et_set_converge(set: *mut et_set, out: *mut et_fr) -> c_int;
    pub fn et_set_converge_f64(set: *mut et_set, alpha: f64, pretrust: *const f64, out: *mut f64,
                               residual_l1: *mut f64) -> c_int;
    pub fn et_set_converge_rational(set: *const et_set, num_le: *mut u8, den_le: *mut u8, stride: usize,
                                    needed: *mut usize) -> c_int;

    // ---- graph engine ----
    pub fn et_graph_build(n: u32, member: *const u8, n_edges: u64, from: *const u32, to: *const u32,
                          value_u32: *const u32, value_fr: *const et_fr, opts: *const et_build_opts,
                          out: *mut *mut et_graph) -> c_int;
    pub fn et_graph_free(g: *mut et_graph);
    pub fn et_graph_info_get(g: *const et_graph, info: *mut et_graph_info) -> c_int;
    pub fn et_plan_build(n: u32, member: *const u8, n_edges: u64, from: *const u32, to: *const u32,
                         value_u32: *const u32, value_fr: *const et_fr, opts: *const et_build_opts,
                         out: *mut *mut et_plan) -> c_int;
    pub fn et_plan_free(p: *mut et_plan);
    pub fn et_plan_info_get(p: *const et_plan, info: *mut et_graph_info) -> c_int;
    pub fn et_plan_arrays_get(p: *const et_plan, out: *mut et_plan_arrays) -> c_int;
    pub fn et_plan_sweep_stats(p: *const et_plan, variant: c_int, out: *mut et_sweep_stats) -> c_int;
    pub fn et_graph_from_plan(p: *const et_plan, opts: *const et_build_opts, out: *mut *mut et_graph) -> c_int;
    pub fn et_graph_converge_fr(g: *mut et_graph, num_iterations: u32, initial_lo: u64, initial_hi: u64,
                                out: *mut et_fr) -> c_int;
    pub fn et_graph_converge_f64(g: *mut et_graph, num_iterations: u32, initial: f64, alpha: f64,
                                 pretrust: *const f64, out: *mut f64, residual_l1: *mut f64) -> c_int;
    pub fn et_graph_init_fr_async(g: *mut et_graph, initial_lo: u64, initial_hi: u64, x: *mut c_void, flags: u32,
                                  stream: *mut c_void) -> c_int;
    pub fn et_graph_init_f64_async(g: *mut et_graph, initial: f64, x: *mut c_void, flags: u32,
                                   stream: *mut c_void) -> c_int;
    pub fn et_graph_step_fr_async(g: *mut et_graph, x: *const c_void, y: *mut c_void, iteration: u32, flags: u32,
                                  stream: *mut c_void) -> c_int;
    pub fn et_graph_step_f64_async(g: *mut et_graph, x: *const c_void, y: *mut c_void, iteration: u32, alpha: f64,
                                   pretrust_pos: *const c_void, flags: u32, stream: *mut c_void) -> c_int;
    pub fn et_graph_step_fr_phase_async(g: *mut et_graph, x: *const c_void, y: *mut c_void, iteration: u32, flags: u32,
                                        phase: u32, stream: *mut c_void) -> c_int;
    pub fn et_graph_step_f64_phase_async(g: *mut et_graph, x: *const c_void, y: *mut c_void, iteration: u32, alpha: f64,
                                         pretrust_pos: *const c_void, flags: u32, phase: u32, stream: *mut c_void) -> c_int;
    pub fn et_graph_exchange_get(g: *const et_graph, out: *mut et_exchange_info) -> c_int;
    pub fn et_plan_exchange_get(p: *const et_plan, out: *mut et_exchange_info, need: *mut *const u32,
                                send: *mut *const u32) -> c_int;
    pub fn et_graph_pack_fr_async(g: *mut et_graph, y: *mut c_void, send: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn et_graph_pack_f64_async(g: *mut et_graph, y: *mut c_void, send: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn et_graph_unpermute_fr_async(g: *mut et_graph, x: *const c_void, out_slots: *mut c_void,
                                       stream: *mut c_void) -> c_int;
    pub fn et_graph_unpermute_f64_async(g: *mut et_graph, x: *const c_void, out_slots: *mut c_void,
                                        stream: *mut c_void) -> c_int;
    pub fn et_graph_permute_f64_host(g: *const et_graph, slots: *const f64, pos: *mut f64) -> c_int;
    pub fn et_graph_stats_fr(g: *mut et_graph, iteration: u32, stream: *mut c_void, out: *mut et_iter_stats) -> c_int;
    pub fn et_graph_stats_f64(g: *mut et_graph, iteration: u32, stream: *mut c_void, out: *mut et_iter_stats) -> c_int;
    pub fn et_graph_run_fr(g: *mut et_graph, num_iterations: u32, initial_lo: u64, initial_hi: u64,
                           stream: *mut c_void, record_events: c_int) -> c_int;
    pub fn et_graph_run_f64(g: *mut et_graph, num_iterations: u32, initial: f64, alpha: f64, stream: *mut c_void,
                            record_events: c_int) -> c_int;
    pub fn et_graph_record_begin(g: *mut et_graph, stream: *mut c_void) -> c_int;
    pub fn et_graph_record_end(g: *mut et_graph, stream: *mut c_void) -> c_int;
    pub fn et_graph_event_times(g: *mut et_graph, total_ms: *mut f32, pairs: *mut u32) -> c_int;
    pub fn et_graph_result_fr(g: *mut et_graph, out: *mut et_fr, stream: *mut c_void) -> c_int;
    pub fn et_graph_result_f64(g: *mut et_graph, out: *mut f64, stream: *mut c_void) -> c_int;

    // ---- batches of small sets ----
    pub fn et_batch_create(n_sets: u32, n: u32, member: *const u8, values: *const u32, device: c_int,
                           out: *mut *mut et_batch) -> c_int;
    pub fn et_batch_free(b: *mut et_batch);
    pub fn et_batch_run_fr_async(b: *mut et_batch, num_iterations: u32, initial_lo: u64, initial_hi: u64,
                                 stream: *mut c_void) -> c_int;
    pub fn et_batch_result_fr(b: *mut et_batch, out: *mut et_fr, status: *mut i32, stream: *mut c_void) -> c_int;
    pub fn et_batch_edges(b: *const et_batch, edges: *mut u64) -> c_int;
    pub fn et_batch_converge_fr(n_sets: u32, n: u32, member: *const u8, values: *const u32, num_iterations: u32,
                                initial_lo: u64, initial_hi: u64, out: *mut et_fr, status: *mut i32) -> c_int;
    /// The same without any device allocation (the batch lives in the library's resident scratch array).
    pub fn et_batch_converge_fr_noalloc(n_sets: u32, n: u32, member: *const u8, values: *const u32, num_iterations: u32,
                                        initial_lo: u64, initial_hi: u64, out: *mut et_fr, status: *mut i32) -> c_int;

    /// Row-sharded converge over the GPUs of one node from one host thread (RCCL all-gather or device copies).
    pub fn et_sharded_create(n: u32, member: *const u8, n_edges: u64, from: *const u32, to: *const u32,
                             value_u32: *const u32, value_fr: *const et_fr, n_devices: u32, devices: *const i32,
                             variants: u32, exchange: u32, n_chunks: u32, out: *mut *mut et_sharded) -> c_int;
    pub fn et_sharded_free(s: *mut et_sharded);
    pub fn et_sharded_exchange_kind(s: *const et_sharded) -> c_int;
    pub fn et_sharded_info_get(s: *const et_sharded, share: u32, info: *mut et_graph_info) -> c_int;
    pub fn et_sharded_converge_fr(s: *mut et_sharded, num_iterations: u32, initial_lo: u64, initial_hi: u64,
                                  out: *mut et_fr) -> c_int;
    pub fn et_sharded_converge_f64(s: *mut et_sharded, num_iterations: u32, initial: f64, alpha: f64,
                                   pretrust: *const f64, out: *mut f64, residual_l1: *mut f64) -> c_int;

    // ---- one share from the attestations of this process's own attesters ----
    pub fn et_dplan_create(n: u32, member: *const u8, n_edges: u64, from: *const u32, to: *const u32,
                           value_u32: *const u32, out: *mut *mut et_dplan) -> c_int;
    pub fn et_dplan_local_stats(p: *const et_dplan, row_sum: *mut u64, in_degree: *mut u32, out_degree: *mut u32,
                                attested: *mut u32, value_max: *mut u32) -> c_int;
    pub fn et_dplan_set_global_stats(p: *mut et_dplan, row_sum: *const u64, in_degree: *const u32,
                                     out_degree: *const u32, attested: *const u32, value_max: u32,
                                     opts: *const et_build_opts) -> c_int;
    pub fn et_dplan_needs(p: *mut et_dplan, share: u32, count: *mut u64, rows: *mut *const u32) -> c_int;
    pub fn et_dplan_add_wanted(p: *mut et_dplan, share: u32, count: u64, rows: *const u32) -> c_int;
    pub fn et_dplan_outgoing(p: *const et_dplan, share: u32, count: *mut u64, src: *mut *const u32,
                             row: *mut *const u32, value: *mut *const u32) -> c_int;
    pub fn et_dplan_add_incoming(p: *mut et_dplan, count: u64, src: *const u32, row: *const u32,
                                 value: *const u32) -> c_int;
    pub fn et_dplan_info_get(p: *const et_dplan, info: *mut et_graph_info) -> c_int;
    pub fn et_dplan_finish(p: *mut et_dplan, out: *mut *mut et_graph) -> c_int;
    pub fn et_dplan_finish_host(p: *mut et_dplan, out: *mut *mut et_plan) -> c_int;
    pub fn et_dplan_free(p: *mut et_dplan);

    pub fn et_synth_graph(kind: u32, n: u32, target_edges: u64, seed: u64, value_max: u32, n_edges: *mut u64,
                          from: *mut u32, to: *mut u32, value: *mut u32) -> c_int;
    pub fn et_synth_graph_range(kind: u32, n: u32, target_edges: u64, seed: u64, value_max: u32, attester_begin: u32,
                                attester_end: u32, n_edges: *mut u64, from: *mut u32, to: *mut u32,
                                value: *mut u32) -> c_int;

    // ---- around converge: threshold, score records, files, public inputs ----
    pub fn et_ratio_decompose(num_le: *const u8, num_len: usize, den_le: *const u8, den_len: usize, num_limbs: u32,
                              power_of_ten: u32, num_limbs_out: *mut et_fr, den_limbs_out: *mut et_fr) -> c_int;
    pub fn et_threshold_check(score_fr: *const et_fr, num_le: *const u8, num_len: usize, den_le: *const u8,
                              den_len: usize, threshold: *const et_fr, num_limbs: u32, power_of_ten: u32,
                              num_neighbours: u32, initial_lo: u64, initial_hi: u64, passed: *mut c_int) -> c_int;
    pub fn et_scalar_from_address(address: *const u8, out: *mut et_fr) -> c_int;
    pub fn et_score_pack(address: *const u8, score_fr: *const et_fr, num_le: *const u8, num_len: usize,
                         den_le: *const u8, den_len: usize, out: *mut et_score) -> c_int;
    pub fn et_attestations_csv_parse(text: *const c_char, len: usize, out: *mut et_attestation, cap: usize,
                                     count: *mut usize) -> c_int;
    pub fn et_attestations_csv_format(recs: *const et_attestation, n: usize, buf: *mut c_char, cap: usize,
                                      needed: *mut usize) -> c_int;
    pub fn et_scores_csv_format(scores: *const et_score, n: usize, buf: *mut c_char, cap: usize,
                                needed: *mut usize) -> c_int;
    pub fn et_public_inputs_pack(participants: *const et_fr, scores: *const et_fr, n: u32, domain: *const et_fr,
                                 opinions_hash: *const et_fr, out: *mut u8, cap: usize) -> c_int;
    pub fn et_public_inputs_unpack(bytes: *const u8, len: usize, n: u32, participants: *mut et_fr,
                                   scores: *mut et_fr, domain: *mut et_fr, opinions_hash: *mut et_fr) -> c_int;
    pub fn et_local_scores(atts: *const et_attestation, attesters: *const u8, n_att: usize, domain: *const u8,
                           num_neighbours: u32, num_iterations: u32, initial_lo: u64, initial_hi: u64,
                           out: *mut et_score, cap: usize, n_participants: *mut usize) -> c_int;

    // ---- opinion validation crypto ----
    pub fn et_poseidon5_permute(input: *const et_fr, out: *mut et_fr) -> c_int;
    pub fn et_poseidon5_sponge(input: *const et_fr, n: usize, out: *mut et_fr) -> c_int;
    pub fn et_keccak256(data: *const u8, len: usize, out: *mut u8) -> c_int;
    pub fn et_attestation_hash(att: *const et_attestation, out: *mut et_fr) -> c_int;
    pub fn et_attestations_recover(atts: *const et_attestation, n: usize, attesters: *mut u8,
                                   public_keys: *mut u8) -> c_int;
    pub fn et_attestations_recover_gpu(atts: *const et_attestation, n: usize, attesters: *mut u8,
                                       public_keys: *mut u8, device: c_int) -> c_int;
    pub fn et_attestation_recover_lane_host(att: *const et_attestation, attester: *mut u8,
                                            public_key: *mut u8) -> c_int;
    pub fn et_secp256k1_public_key(secret_le: *const u8, public_key: *mut u8, address: *mut u8) -> c_int;
    pub fn et_secp256k1_sign(secret_le: *const u8, msg_hash: *const et_fr, nonce_le: *const u8, sig_r: *mut u8,
                             sig_s: *mut u8, rec_id: *mut u8) -> c_int;
    pub fn et_opinion_validate(public_key: *const u8, atts: *const et_attestation, present: *const u8,
                               set_addresses: *const u8, num_neighbours: u32, domain: *const u8,
                               scores: *mut et_fr, op_hash: *mut et_fr) -> c_int;
    pub fn et_et_setup(atts: *const et_attestation, n_att: usize, domain: *const u8, num_neighbours: u32,
                       num_iterations: u32, initial_lo: u64, initial_hi: u64, out: *mut et_score, cap: usize,
                       n_participants: *mut usize, public_inputs: *mut u8, public_inputs_cap: usize) -> c_int;
    pub fn et_scores_from_attestations(atts: *const et_attestation, n_att: usize, domain: *const u8,
                                       num_iterations: u32, initial_lo: u64, initial_hi: u64, addresses: *mut u8,
                                       scores_fr: *mut et_fr, scores_f64: *mut f64, cap: usize,
                                       n_participants: *mut usize) -> c_int;

    pub fn et_dev_fr_mul(count: u32, a: *const et_fr, b: *const et_fr, out: *mut et_fr) -> c_int;
}
