//! `EigenTrustSet`-shaped wrapper over the HIP engine: same method names, argument meaning and
//! panics as eigentrust-zk/src/circuits/dynamic_sets/native.rs:165-392, so that
//! eigentrust/src/lib.rs:422-440 can switch by changing one `use`.
//!
//! Signature cryptography stays where it is: `update_op` here takes the score row that
//! `Opinion::validate` (opinion/native.rs:63-109) returns; `update_op_signed` hands the records to
//! the library's own validation instead.
//!
//! NOT COMPILED in the image this repository is built in (no Rust toolchain).

use eigentrust_hip_sys as sys;
use num_bigint::{BigInt, Sign};
use num_rational::BigRational;
use std::ffi::CStr;

/// What the wrapper needs from the scalar type; for `halo2curves::bn256::Fr`:
/// `to_le_bytes = |f| f.to_repr()` and `from_le_bytes = |b| Fr::from_repr(b).unwrap()`.
pub trait ScalarBytes: Copy {
    fn to_le_bytes(&self) -> [u8; 32];
    fn from_le_bytes(bytes: [u8; 32]) -> Self;
}

fn fr<S: ScalarBytes>(s: &S) -> sys::et_fr {
    sys::et_fr { b: s.to_le_bytes() }
}

/// The reference panics where the library returns a code (native.rs:178,181,192,289,334).
fn check(rc: i32) {
    if rc != sys::ET_OK {
        let what = unsafe { CStr::from_ptr(sys::et_strerror(rc)) }.to_string_lossy().into_owned();
        let detail = unsafe { CStr::from_ptr(sys::et_last_error_message()) }.to_string_lossy().into_owned();
        panic!("eigentrust-hip: {} ({}) {}", what, rc, detail);
    }
}

pub struct EigenTrustSet<S: ScalarBytes, const NUM_NEIGHBOURS: usize, const NUM_ITERATIONS: usize, const INITIAL_SCORE: u128> {
    handle: *mut sys::et_set,
    _scalar: std::marker::PhantomData<S>,
}

impl<S: ScalarBytes, const NUM_NEIGHBOURS: usize, const NUM_ITERATIONS: usize, const INITIAL_SCORE: u128>
    EigenTrustSet<S, NUM_NEIGHBOURS, NUM_ITERATIONS, INITIAL_SCORE>
{
    /// native.rs:165-172
    pub fn new(domain: S) -> Self {
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            sys::et_set_new(NUM_NEIGHBOURS as u32, NUM_ITERATIONS as u32, INITIAL_SCORE as u64,
                            (INITIAL_SCORE >> 64) as u64, &fr(&domain), &mut handle)
        });
        Self { handle, _scalar: std::marker::PhantomData }
    }

    /// native.rs:175-186
    pub fn add_member(&mut self, addr: S) {
        check(unsafe { sys::et_set_add_member(self.handle, &fr(&addr)) });
    }

    /// native.rs:189-198
    pub fn remove_member(&mut self, addr: S) {
        check(unsafe { sys::et_set_remove_member(self.handle, &fr(&addr)) });
    }

    /// native.rs:201-231 after validation: `scores` is the row `Opinion::validate` returned for `from`.
    pub fn update_op(&mut self, from: S, scores: &[S]) {
        assert_eq!(scores.len(), NUM_NEIGHBOURS);
        let raw: Vec<sys::et_fr> = scores.iter().map(fr).collect();
        check(unsafe { sys::et_set_update_op(self.handle, &fr(&from), raw.as_ptr()) });
    }

    /// native.rs:201-231 including `Opinion::validate`: `op[i] == None` is the reference's
    /// `empty_with_about` (native.rs:209-219).  Returns the opinion hash.
    pub fn update_op_signed(&mut self, public_key: &[u8; 64], from: S, op: &[Option<sys::et_attestation>],
                            domain: &[u8; 20]) -> S {
        assert_eq!(op.len(), NUM_NEIGHBOURS);
        let (addrs, _) = self.members_raw();
        let mut set_addresses = vec![0u8; NUM_NEIGHBOURS * 20];
        for (i, a) in addrs.iter().enumerate() {
            // an address scalar is the 20 address bytes, big-endian, as an integer (eth.rs:77-95)
            for k in 0..20 {
                set_addresses[i * 20 + k] = a.b[19 - k];
            }
        }
        let blank = sys::et_attestation { about: [0; 20], domain: [0; 20], value: 0, message: [0; 32],
                                          sig_r: [0; 32], sig_s: [0; 32], rec_id: 0 };
        let atts: Vec<sys::et_attestation> = op.iter().map(|a| a.unwrap_or(blank)).collect();
        let present: Vec<u8> = op.iter().map(|a| a.is_some() as u8).collect();
        let mut scores = vec![sys::et_fr::default(); NUM_NEIGHBOURS];
        let mut op_hash = sys::et_fr::default();
        check(unsafe {
            sys::et_opinion_validate(public_key.as_ptr(), atts.as_ptr(), present.as_ptr(), set_addresses.as_ptr(),
                                     NUM_NEIGHBOURS as u32, domain.as_ptr(), scores.as_mut_ptr(), &mut op_hash)
        });
        check(unsafe { sys::et_set_update_op(self.handle, &fr(&from), scores.as_ptr()) });
        S::from_le_bytes(op_hash.b)
    }

    /// native.rs:286-337: filter, normalise, NUM_ITERATIONS exact iterations on the GPU; panics on
    /// fewer than two members (:289) and on a changed score sum (:334).
    pub fn converge(&self) -> Vec<S> {
        let mut out = vec![sys::et_fr::default(); NUM_NEIGHBOURS];
        check(unsafe { sys::et_set_converge(self.handle, out.as_mut_ptr()) });
        out.iter().map(|e| S::from_le_bytes(e.b)).collect()
    }

    /// The f64 iteration of the same update, with optional damping towards `pretrust`.
    pub fn converge_f64(&self, alpha: f64, pretrust: Option<&[f64]>) -> (Vec<f64>, f64) {
        let mut out = vec![0f64; NUM_NEIGHBOURS];
        let mut residual = 0f64;
        let p = pretrust.map_or(std::ptr::null(), |p| p.as_ptr());
        check(unsafe { sys::et_set_converge_f64(self.handle, alpha, p, out.as_mut_ptr(), &mut residual) });
        (out, residual)
    }

    /// native.rs:340-392 (host side, exact fractions).
    pub fn converge_rational(&self) -> Vec<BigRational> {
        let mut needed = 0usize;
        let rc = unsafe {
            sys::et_set_converge_rational(self.handle, std::ptr::null_mut(), std::ptr::null_mut(), 0, &mut needed)
        };
        if rc != sys::ET_E_BUFFER_TOO_SMALL {
            check(rc);
        }
        let stride = needed.max(1);
        let mut num = vec![0u8; stride * NUM_NEIGHBOURS];
        let mut den = vec![0u8; stride * NUM_NEIGHBOURS];
        check(unsafe {
            sys::et_set_converge_rational(self.handle, num.as_mut_ptr(), den.as_mut_ptr(), stride, &mut needed)
        });
        (0..NUM_NEIGHBOURS)
            .map(|i| {
                let n = BigInt::from_bytes_le(Sign::Plus, &num[i * stride..(i + 1) * stride]);
                let d = BigInt::from_bytes_le(Sign::Plus, &den[i * stride..(i + 1) * stride]);
                BigRational::new(n, d)
            })
            .collect()
    }

    fn members_raw(&self) -> (Vec<sys::et_fr>, Vec<sys::et_fr>) {
        let mut addrs = vec![sys::et_fr::default(); NUM_NEIGHBOURS];
        let mut scores = vec![sys::et_fr::default(); NUM_NEIGHBOURS];
        check(unsafe { sys::et_set_members(self.handle, addrs.as_mut_ptr(), scores.as_mut_ptr()) });
        (addrs, scores)
    }

    /// The `set` field of the reference (native.rs:120): (address, initial score) per slot.
    pub fn set(&self) -> Vec<(S, S)> {
        let (a, s) = self.members_raw();
        a.iter().zip(s.iter()).map(|(a, s)| (S::from_le_bytes(a.b), S::from_le_bytes(s.b))).collect()
    }
}

impl<S: ScalarBytes, const NUM_NEIGHBOURS: usize, const NUM_ITERATIONS: usize, const INITIAL_SCORE: u128> Drop
    for EigenTrustSet<S, NUM_NEIGHBOURS, NUM_ITERATIONS, INITIAL_SCORE>
{
    fn drop(&mut self) {
        unsafe { sys::et_set_free(self.handle) };
    }
}

/// The production alias (circuits/mod.rs:39-45,111-123): 4 slots, 20 iterations, INITIAL_SCORE 1000.
pub type NativeEigenTrust4<S> = EigenTrustSet<S, 4, 20, 1000>;
