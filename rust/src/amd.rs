//! needletail on MI355X: FFI binding of `libneedletail_amd.so` (include/needletail_amd.h) and the adapters a
//! `cfg(feature = "amd")` build of needletail would use.
//!
//! NOT COMPILED IN THIS REPOSITORY: the build image has no rustc / cargo.  The declarations below
//! are generated from the header by tools/gen_rust_ffi.py (tests/test_abi.py diffs the block against the generator: types and constness); the adapters follow the reference's
//! own types: `Sequence::canonical_kmers` / `bit_kmers` (src/sequence.rs:237-252), `CanonicalKmers::next`
//! (src/kmer.rs:84-129), `BitNuclKmer::next` (src/bitkmer.rs:80-109), `FastxReader::next` (src/parser/utils.rs:119-130).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const NTK_OK: c_int = 0;
pub const NTK_ERR_CAPACITY: c_int = 5;
pub const NTK_EOF: c_int = 100;
pub const NTK_PATH_BYTES_CANONICAL: u32 = 0;
pub const NTK_PATH_BITS: u32 = 1;
pub const NTK_PATH_BITS_CANONICAL: u32 = 2;
pub const NTK_PRE_NONE: u32 = 0;
pub const NTK_PRE_STRIP_RETURNS: u32 = 1;
pub const NTK_PRE_NORMALIZE: u32 = 2;
pub const NTK_PRE_NORMALIZE_IUPAC: u32 = 3;
pub const NTK_HIST_BINS: usize = 4096;
/// ntk_params.flags bit 16: the reduce call zeroes the accumulators first, inside its own kernel launch.
pub const NTK_FLAG_RESET: u32 = 1 << 16;
pub const NTK_COMM_ID_BYTES: usize = 128;

#[repr(C)] #[derive(Clone, Copy)]
pub struct NtkParams { pub k: u32, pub path: u32, pub pre: u32, pub flags: u32 }
/// flags = w | (quality_cutoff << 8): w > 0 folds windowed minimizers, a cutoff masks low-quality FASTQ bases first
pub const fn ntk_flags(window_w: u32, quality_cutoff: u32) -> u32 { (window_w & 0xFF) | ((quality_cutoff & 0xFF) << 8) }
#[repr(C)]
pub struct NtkResult { pub n_total: u64, pub n_fwd: u64, pub n_rc: u64, pub sum: u64, pub xr: u64, pub hist: [u64; NTK_HIST_BINS], pub n_undigested: u64 }
#[repr(C)]
pub struct NtkRecord {
    pub id: *const u8, pub id_len: u64, pub seq: *const u8, pub seq_len: u64, pub qual: *const u8, pub qual_len: u64,
    pub format: u32, pub line_ending: u32, pub line: u64, pub num_bases: u64, pub byte: u64,
}
/// ntk_gunzip: which route inflated the file (1 block gzip, 2 speculative parallel inflate of an ordinary stream, 3 sequential) and its timings
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct NtkGunzipInfo {
    pub route: u32, pub threads: u32, pub chunks: u32, pub chunks_dropped: u32, pub members: u32, pub streamed: u32,
    pub search_s: f64, pub decode_s: f64, pub decode_busy_s: f64, pub crc_s: f64, pub marker_symbols: u64,
    // ABI 4: streamed runs of ntk_scan_file_parallel (ntk_scan_file_info)
    pub chunks_deferred: u32, pub parse_threads: u32, pub peak_backlog_bytes: u64, pub text_bytes: u64,
    pub first_batch_s: f64, pub total_s: f64, pub resolve_busy_s: f64,
}
pub enum NtkCtx {}
pub enum NtkBatch {}
pub enum NtkReader {}
pub enum NtkComm {}

extern "C" {
    pub fn ntk_strerror(status: c_int) -> *const c_char;
    pub fn ntk_last_hip_error() -> c_int;
    pub fn ntk_last_rccl_error() -> c_int;
    pub fn ntk_abi_version() -> c_int;
    pub fn ntk_device_count(n: *mut c_int) -> c_int;
    pub fn ntk_ctx_create(device: c_int, out: *mut *mut NtkCtx) -> c_int;
    pub fn ntk_ctx_create_on_stream(device: c_int, hip_stream: *mut c_void, out: *mut *mut NtkCtx) -> c_int;
    pub fn ntk_ctx_destroy(ctx: *mut NtkCtx);
    pub fn ntk_ctx_synchronize(ctx: *mut NtkCtx) -> c_int;
    pub fn ntk_ctx_set_launch(ctx: *mut NtkCtx, blocks: c_int, threads_per_block: c_int) -> c_int;
    pub fn ntk_ctx_set_option(ctx: *mut NtkCtx, option: c_int, value: u64) -> c_int;
    pub fn ntk_ctx_get_option(ctx: *mut NtkCtx, option: c_int, value: *mut u64) -> c_int;
    pub fn ntk_ctx_enable_timing(ctx: *mut NtkCtx, on: c_int) -> c_int;
    pub fn ntk_ctx_scan_time_ms(ctx: *mut NtkCtx, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn ntk_comm_init_all(ctxs: *const *mut NtkCtx, n: c_int, out: *mut *mut NtkComm) -> c_int;
    pub fn ntk_comm_unique_id(id: *mut u8) -> c_int;
    pub fn ntk_comm_init_rank(ctx: *mut NtkCtx, n_ranks: c_int, rank: c_int, id: *const u8, out: *mut *mut NtkComm) -> c_int;
    pub fn ntk_comm_size(comm: *const NtkComm) -> c_int;
    pub fn ntk_allreduce_accumulators(comm: *mut NtkComm) -> c_int;
    pub fn ntk_comm_allreduce_time_ms(comm: *mut NtkComm, total_ms: *mut f64, calls: *mut u64) -> c_int;
    pub fn ntk_comm_destroy(comm: *mut NtkComm);
    pub fn ntk_accum_reset(ctx: *mut NtkCtx) -> c_int;
    pub fn ntk_reduce_device(ctx: *mut NtkCtx, d_seq: *const u8, n_bytes: u64, p: *const NtkParams) -> c_int;
    pub fn ntk_reduce_device_quality(ctx: *mut NtkCtx, d_seq: *const u8, d_qual: *const u8, n_bytes: u64, p: *const NtkParams) -> c_int;
    pub fn ntk_accum_read(ctx: *mut NtkCtx, out: *mut NtkResult) -> c_int;
    pub fn ntk_accum_device_ptr(ctx: *mut NtkCtx, d_words: *mut *mut u64) -> c_int;
    pub fn ntk_accum_bind_device(ctx: *mut NtkCtx, d_words: *mut u64) -> c_int;
    pub fn ntk_materialize_device(ctx: *mut NtkCtx, d_seq: *const u8, n_bytes: u64, p: *const NtkParams, d_values: *mut u64, d_valid16: *mut u16, d_rc16: *mut u16) -> c_int;
    pub fn ntk_materialize_device_quality(ctx: *mut NtkCtx, d_seq: *const u8, d_qual: *const u8, n_bytes: u64, p: *const NtkParams, d_values: *mut u64, d_valid16: *mut u16, d_rc16: *mut u16) -> c_int;
    pub fn ntk_batch_acquire(ctx: *mut NtkCtx, max_bytes: u64, max_records: u64, out: *mut *mut NtkBatch) -> c_int;
    pub fn ntk_batch_append(b: *mut NtkBatch, seq: *const u8, n: u64, pre: u32) -> c_int;
    pub fn ntk_batch_append_quality(b: *mut NtkBatch, seq: *const u8, qual: *const u8, n: u64, pre: u32, cutoff: u32) -> c_int;
    pub fn ntk_batch_buffers(b: *mut NtkBatch, seq: *mut *mut u8, offsets: *mut *mut u64, n_bytes: *mut u64, n_records: *mut u64) -> c_int;
    pub fn ntk_batch_submit(ctx: *mut NtkCtx, b: *mut NtkBatch, p: *const NtkParams) -> c_int;
    pub fn ntk_batch_wait(ctx: *mut NtkCtx, b: *mut NtkBatch) -> c_int;
    pub fn ntk_batch_release(ctx: *mut NtkCtx, b: *mut NtkBatch);
    pub fn ntk_reader_open_file(path: *const c_char, out: *mut *mut NtkReader) -> c_int;
    pub fn ntk_reader_open_memory(data: *const u8, n: u64, out: *mut *mut NtkReader) -> c_int;
    pub fn ntk_reader_next(r: *mut NtkReader, rec: *mut NtkRecord) -> c_int;
    pub fn ntk_reader_error(r: *mut NtkReader, kind: *mut c_int, line: *mut u64, msg: *mut c_char, msg_cap: u64, id: *mut c_char, id_cap: u64) -> c_int;
    pub fn ntk_reader_position(r: *mut NtkReader, line: *mut u64, byte: *mut u64, ending: *mut c_int) -> c_int;
    pub fn ntk_reader_close(r: *mut NtkReader);
    pub fn ntk_scan_reader(ctx: *mut NtkCtx, r: *mut NtkReader, p: *const NtkParams, batch_bytes: u64, n_batches: u32, n_records: *mut u64, n_bases: *mut u64) -> c_int;
    pub fn ntk_fastx_split_points(data: *const u8, n: u64, n_pieces: u32, cuts: *mut u64) -> c_int;
    pub fn ntk_scan_buffer_parallel(ctx: *mut NtkCtx, data: *const u8, n: u64, p: *const NtkParams, batch_bytes: u64, n_threads: u32, n_records: *mut u64, n_bases: *mut u64) -> c_int;
    pub fn ntk_scan_file_parallel(ctx: *mut NtkCtx, path: *const c_char, p: *const NtkParams, batch_bytes: u64, n_threads: u32, n_records: *mut u64, n_bases: *mut u64) -> c_int;
    pub fn ntk_gunzip(gz: *const u8, n: u64, n_threads: u32, out: *mut *mut u8, out_n: *mut u64, info: *mut NtkGunzipInfo) -> c_int;
    pub fn ntk_gunzip_free(out: *mut u8, out_n: u64);
    pub fn ntk_scan_file_info(info: *mut NtkGunzipInfo) -> c_int;
    pub fn ntk_normalize(ctx: *mut NtkCtx, seq: *const u8, n: u64, allow_iupac: c_int, out: *mut u8, out_len: *mut u64, changed: *mut c_int) -> c_int;
    pub fn ntk_strip_returns(ctx: *mut NtkCtx, seq: *const u8, n: u64, out: *mut u8, out_len: *mut u64, borrowed: *mut c_int) -> c_int;
    pub fn ntk_reverse_complement(ctx: *mut NtkCtx, seq: *const u8, n: u64, out: *mut u8) -> c_int;
    pub fn ntk_canonical_kmers(ctx: *mut NtkCtx, seq: *const u8, n: u64, k: u32, pos_out: *mut u64, is_rc_out: *mut u8, cap: u64, count: *mut u64) -> c_int;
    pub fn ntk_bit_kmers(ctx: *mut NtkCtx, seq: *const u8, n: u64, k: u32, canonical: c_int, pos_out: *mut u64, val_out: *mut u64, was_rc_out: *mut u8, cap: u64, count: *mut u64) -> c_int;
    pub fn ntk_canonical_kmers_batch(ctx: *mut NtkCtx, seq: *const u8, offsets: *const u64, n_records: u64, k: u32, counts: *mut u64, pos_out: *mut u64, is_rc_out: *mut u8, cap: u64, total: *mut u64) -> c_int;
    pub fn ntk_bit_kmers_batch(ctx: *mut NtkCtx, seq: *const u8, offsets: *const u64, n_records: u64, k: u32, canonical: c_int, counts: *mut u64, pos_out: *mut u64, val_out: *mut u64, was_rc_out: *mut u8, cap: u64, total: *mut u64) -> c_int;
    pub fn ntk_pinned_alloc(bytes: u64, out: *mut *mut c_void) -> c_int;
    pub fn ntk_canonical_kmers_batch_planes(ctx: *mut NtkCtx, seq: *const u8, offsets: *const u64, n_records: u64, k: u32, rec_bit: *mut u64, valid16: *mut u16, rc16: *mut u16, cap_words: u64, n_words: *mut u64, total: *mut u64) -> c_int;
    pub fn ntk_bit_kmers_batch_planes(ctx: *mut NtkCtx, seq: *const u8, offsets: *const u64, n_records: u64, k: u32, canonical: c_int, rec_bit: *mut u64, valid16: *mut u16, rc16: *mut u16, values: *mut u64, cap_words: u64, n_words: *mut u64, total: *mut u64) -> c_int;
    pub fn ntk_ctx_trim(ctx: *mut NtkCtx) -> c_int;
    pub fn ntk_pinned_free(p: *mut c_void);
    pub fn ntk_minimizers_reduce_device(ctx: *mut NtkCtx, d_seq: *const u8, n_bytes: u64, p: *const NtkParams, w: u32) -> c_int;
    pub fn ntk_minimizer(ctx: *mut NtkCtx, seq: *const u8, n: u64, m: u32, out: *mut u8) -> c_int;
    pub fn ntk_minimizer_batch(ctx: *mut NtkCtx, seq: *const u8, offsets: *const u64, n_records: u64, m: u32, out: *mut u8, pos_out: *mut u64, is_rc_out: *mut u8, bad_record: *mut u64) -> c_int;
    pub fn ntk_canonical(ctx: *mut NtkCtx, seq: *const u8, n: u64, out: *mut u8, was_rc: *mut c_int) -> c_int;
    pub fn ntk_bit_minimizers(ctx: *mut NtkCtx, values: *const u64, n: u64, k: u32, m: u32, out: *mut u64) -> c_int;
    pub fn ntk_bit_canonical(ctx: *mut NtkCtx, values: *const u64, n: u64, k: u32, canonical: c_int, out: *mut u64, was_rc_out: *mut u8) -> c_int;
    pub fn ntk_quality_mask(ctx: *mut NtkCtx, seq: *const u8, qual: *const u8, n: u64, score: u8, out: *mut u8) -> c_int;
    pub fn ntk_synth_reads_device(ctx: *mut NtkCtx, seed: u64, first_read: u64, n_reads: u64, read_len: u32, n_per_1024: u32, d_out: *mut u8) -> c_int;
    pub fn ntk_reverse_complement_records_device(ctx: *mut NtkCtx, d_in: *const u8, d_out: *mut u8, n_records: u64, record_len: u32, stride: u32) -> c_int;
}

#[derive(Debug)]
pub struct AmdError(pub c_int);
fn check(rc: c_int) -> Result<(), AmdError> { if rc == NTK_OK { Ok(()) } else { Err(AmdError(rc)) } }

/// One context per (thread, device): `Send`, not `Sync`, like the reference's readers (src/parser/utils.rs:119).
pub struct AmdContext(pub *mut NtkCtx);
unsafe impl Send for AmdContext {}
impl AmdContext {
    pub fn new(device: i32) -> Result<Self, AmdError> {
        let mut p = std::ptr::null_mut();
        check(unsafe { ntk_ctx_create(device, &mut p) })?;
        Ok(AmdContext(p))
    }
    pub fn result(&self) -> Result<Box<NtkResult>, AmdError> {
        let mut r: Box<NtkResult> = unsafe { Box::new(std::mem::zeroed()) };
        check(unsafe { ntk_accum_read(self.0, &mut *r) })?;
        Ok(r)
    }
}
impl Drop for AmdContext { fn drop(&mut self) { unsafe { ntk_ctx_destroy(self.0) } } }

/// The items of `Sequence::canonical_kmers(k, &rc)` for a whole batch of records, fetched by ONE call
/// (ntk_canonical_kmers_batch); `iter(i)` is a drop-in for the iterator the trait method returns for record i:
/// same Item, same order, the slice drawn from `buffer` or from the caller's `rc` exactly as src/kmer.rs:121-128.
pub struct AmdCanonicalKmersBatch { k: usize, counts: Vec<u64>, starts: Vec<usize>, pos: Vec<u64>, is_rc: Vec<u8> }
impl AmdCanonicalKmersBatch {
    pub fn new(ctx: &AmdContext, records: &[&[u8]], k: u8) -> Result<Self, AmdError> {
        let mut seq = Vec::with_capacity(records.iter().map(|r| r.len()).sum());
        let mut offsets = Vec::with_capacity(records.len() + 1);
        offsets.push(0u64);
        let mut cap = 0u64;
        for r in records {
            seq.extend_from_slice(r);
            offsets.push(seq.len() as u64);
            cap += (r.len() as u64 + 1).saturating_sub(k as u64);
        }
        let (mut counts, mut pos, mut is_rc) = (vec![0u64; records.len()], vec![0u64; cap as usize], vec![0u8; cap as usize]);
        let mut total = 0u64;
        check(unsafe { ntk_canonical_kmers_batch(ctx.0, seq.as_ptr(), offsets.as_ptr(), records.len() as u64, k as u32,
                                                  counts.as_mut_ptr(), pos.as_mut_ptr(), is_rc.as_mut_ptr(), cap, &mut total) })?;
        pos.truncate(total as usize); is_rc.truncate(total as usize);
        let mut starts = Vec::with_capacity(records.len() + 1);
        let mut run = 0usize;
        for c in &counts { starts.push(run); run += *c as usize; }
        starts.push(run);
        Ok(Self { k: k as usize, counts, starts, pos, is_rc })
    }
    pub fn iter<'a>(&'a self, i: usize, buffer: &'a [u8], rc: &'a [u8]) -> impl Iterator<Item = (usize, &'a [u8], bool)> + 'a {
        let (k, n) = (self.k, rc.len());
        (self.starts[i]..self.starts[i + 1]).map(move |j| {
            let (p, f) = (self.pos[j] as usize, self.is_rc[j] != 0);
            if f { (p, &rc[n - p - k..n - p], true) } else { (p, &buffer[p..p + k], false) }
        })
    }
}

/// The same items as two BIT PLANES (ntk_canonical_kmers_batch_planes): per window start "emitted" and "is_rc" - a quarter of a byte
/// per sequence byte crosses PCIe instead of nine bytes per item, and the records are uploaded as they lie in one contiguous buffer.
/// `iter(i, buffer, rc)` walks record i's bits and yields what `CanonicalKmers` yields (src/kmer.rs:114-129).
pub struct AmdCanonicalKmersPlanes { k: usize, lens: Vec<usize>, rec_bit: Vec<u64>, valid16: Vec<u16>, rc16: Vec<u16>, pub total: u64 }
impl AmdCanonicalKmersPlanes {
    /// `seq` + `offsets` (n + 1 entries): record i = seq[offsets[i]..offsets[i + 1]] - e.g. the reader's own buffer, no copy
    pub fn new(ctx: &AmdContext, seq: &[u8], offsets: &[u64], k: u8) -> Result<Self, AmdError> {
        if offsets.is_empty() {   // no offsets at all = zero records (n + 1 entries describe n records)
            return Ok(Self { k: k as usize, lens: Vec::new(), rec_bit: Vec::new(), valid16: Vec::new(), rc16: Vec::new(), total: 0 });
        }
        let n = offsets.len() - 1;
        let cap = (offsets[n] - offsets[0]) / 16 + n as u64 + 1;
        let (mut rec_bit, mut valid16, mut rc16) = (vec![0u64; n + 1], vec![0u16; cap as usize], vec![0u16; cap as usize]);
        let (mut words, mut total) = (0u64, 0u64);
        check(unsafe { ntk_canonical_kmers_batch_planes(ctx.0, seq.as_ptr(), offsets.as_ptr(), n as u64, k as u32, rec_bit.as_mut_ptr(),
                                                         valid16.as_mut_ptr(), rc16.as_mut_ptr(), cap, &mut words, &mut total) })?;
        valid16.truncate(words as usize); rc16.truncate(words as usize);
        let lens = (0..n).map(|i| (offsets[i + 1] - offsets[i]) as usize).collect();
        Ok(Self { k: k as usize, lens, rec_bit, valid16, rc16, total })
    }
    #[inline] fn bit(plane: &[u16], b: u64) -> bool { (plane[(b >> 4) as usize] >> (15 - (b & 15))) & 1 != 0 }
    pub fn iter<'a>(&'a self, i: usize, buffer: &'a [u8], rc: &'a [u8]) -> impl Iterator<Item = (usize, &'a [u8], bool)> + 'a {
        let (k, n, b0) = (self.k, rc.len(), self.rec_bit[i]);
        let windows = (self.lens[i] + 1).saturating_sub(k);
        (0..windows).filter(move |&p| Self::bit(&self.valid16, b0 + p as u64)).map(move |p| {
            if Self::bit(&self.rc16, b0 + p as u64) { (p, &rc[n - p - k..n - p], true) } else { (p, &buffer[p..p + k], false) }
        })
    }
}

/// `Sequence::bit_kmers(k, canonical)` for a whole batch as bit planes + dense values (ntk_bit_kmers_batch_planes): 8.25 bytes per position
/// come back instead of 17 per item, no compaction pass.  `iter(i)` yields what `BitNuclKmer` yields for record i (src/bitkmer.rs:97-108).
pub struct AmdBitKmersPlanes { k: u8, lens: Vec<usize>, rec_bit: Vec<u64>, valid16: Vec<u16>, rc16: Vec<u16>, values: Vec<u64>, pub total: u64 }
impl AmdBitKmersPlanes {
    pub fn new(ctx: &AmdContext, seq: &[u8], offsets: &[u64], k: u8, canonical: bool) -> Result<Self, AmdError> {
        if offsets.is_empty() {
            return Ok(Self { k, lens: Vec::new(), rec_bit: Vec::new(), valid16: Vec::new(), rc16: Vec::new(), values: Vec::new(), total: 0 });
        }
        let n = offsets.len() - 1;
        let cap = (offsets[n] - offsets[0]) / 16 + n as u64 + 1;
        let (mut rec_bit, mut valid16, mut rc16, mut values) =
            (vec![0u64; n + 1], vec![0u16; cap as usize], vec![0u16; cap as usize], vec![0u64; cap as usize * 16]);
        let (mut words, mut total) = (0u64, 0u64);
        check(unsafe { ntk_bit_kmers_batch_planes(ctx.0, seq.as_ptr(), offsets.as_ptr(), n as u64, k as u32, canonical as c_int, rec_bit.as_mut_ptr(),
                                                   valid16.as_mut_ptr(), rc16.as_mut_ptr(), values.as_mut_ptr(), cap, &mut words, &mut total) })?;
        valid16.truncate(words as usize); rc16.truncate(words as usize); values.truncate(words as usize * 16);
        let lens = (0..n).map(|i| (offsets[i + 1] - offsets[i]) as usize).collect();
        Ok(Self { k, lens, rec_bit, valid16, rc16, values, total })
    }
    #[inline] fn bit(plane: &[u16], b: u64) -> bool { (plane[(b >> 4) as usize] >> (15 - (b & 15))) & 1 != 0 }
    pub fn iter<'a>(&'a self, i: usize) -> impl Iterator<Item = (usize, (u64, u8), bool)> + 'a {
        let (k, b0) = (self.k, self.rec_bit[i]);
        let windows = (self.lens[i] + 1).saturating_sub(k as usize);
        (0..windows).filter(move |&p| Self::bit(&self.valid16, b0 + p as u64))
            .map(move |p| (p, (self.values[(b0 + p as u64) as usize], k), Self::bit(&self.rc16, b0 + p as u64)))
    }
}

/// `Sequence::bit_kmers(k, canonical)` for a whole batch (ntk_bit_kmers_batch): Item = (pos, (value, k), was_rc).
pub struct AmdBitKmersBatch { k: u8, starts: Vec<usize>, pos: Vec<u64>, val: Vec<u64>, was_rc: Vec<u8> }
impl AmdBitKmersBatch {
    pub fn new(ctx: &AmdContext, records: &[&[u8]], k: u8, canonical: bool) -> Result<Self, AmdError> {
        let mut seq = Vec::new();
        let mut offsets = vec![0u64];
        let mut cap = 0u64;
        for r in records { seq.extend_from_slice(r); offsets.push(seq.len() as u64); cap += (r.len() as u64 + 1).saturating_sub(k as u64); }
        let mut counts = vec![0u64; records.len()];
        let (mut pos, mut val, mut was_rc) = (vec![0u64; cap as usize], vec![0u64; cap as usize], vec![0u8; cap as usize]);
        let mut total = 0u64;
        check(unsafe { ntk_bit_kmers_batch(ctx.0, seq.as_ptr(), offsets.as_ptr(), records.len() as u64, k as u32, canonical as c_int,
                                            counts.as_mut_ptr(), pos.as_mut_ptr(), val.as_mut_ptr(), was_rc.as_mut_ptr(), cap, &mut total) })?;
        let mut starts = Vec::with_capacity(records.len() + 1);
        let mut run = 0usize;
        for c in &counts { starts.push(run); run += *c as usize; }
        starts.push(run);
        Ok(Self { k, starts, pos, val, was_rc })
    }
    pub fn iter(&self, i: usize) -> impl Iterator<Item = (usize, (u64, u8), bool)> + '_ {
        (self.starts[i]..self.starts[i + 1]).map(move |j| (self.pos[j] as usize, (self.val[j], self.k), self.was_rc[j] != 0))
    }
}

/// `sequence::minimizer(seq, length)` (reference src/sequence.rs:139-152) for every record of a reader batch in one call
/// (`ntk_minimizer_batch`): `mins[r * length .. (r + 1) * length]` = record r's minimizer, `pos[r]` the winning window's start on its
/// strand's string, `is_rc[r]` its strand.  A record shorter than `length` is an error (the reference panics there).
pub fn minimizer_batch(ctx: &AmdContext, seq: &[u8], offsets: &[u64], length: u32) -> Result<(Vec<u8>, Vec<u64>, Vec<u8>), AmdError> {
    let n = offsets.len().saturating_sub(1);
    let (mut mins, mut pos, mut is_rc) = (vec![0u8; n * length as usize], vec![0u64; n], vec![0u8; n]);
    let mut bad = 0u64;
    check(unsafe { ntk_minimizer_batch(ctx.0, seq.as_ptr(), offsets.as_ptr(), n as u64, length, mins.as_mut_ptr(), pos.as_mut_ptr(), is_rc.as_mut_ptr(), &mut bad) })?;
    Ok((mins, pos, is_rc))
}

/// The fast path: what the README loop (src/lib.rs:15-35) becomes.  Records go into pinned batches; a full batch is
/// submitted (hipMemcpyAsync on the copy stream + scan kernels) while the next one fills; results stay on the device
/// until `finish`.
pub struct AmdBatchScanner<'c> { ctx: &'c AmdContext, params: NtkParams, batches: [*mut NtkBatch; 2], cur: usize, in_flight: [bool; 2] }
impl<'c> AmdBatchScanner<'c> {
    pub fn new(ctx: &'c AmdContext, k: u8, path: u32, pre: u32, batch_bytes: u64) -> Result<Self, AmdError> {
        let mut b = [std::ptr::null_mut(); 2];
        for slot in b.iter_mut() { check(unsafe { ntk_batch_acquire(ctx.0, batch_bytes, batch_bytes / 32 + 16, slot) })?; }
        check(unsafe { ntk_accum_reset(ctx.0) })?;
        Ok(Self { ctx, params: NtkParams { k: k as u32, path, pre, flags: 0 }, batches: b, cur: 0, in_flight: [false; 2] })
    }
    /// `rec.sequence()` of one parsed record (src/parser/record.rs:181-185)
    pub fn push(&mut self, sequence: &[u8]) -> Result<(), AmdError> {
        let rc = unsafe { ntk_batch_append(self.batches[self.cur], sequence.as_ptr(), sequence.len() as u64, self.params.pre) };
        if rc != NTK_ERR_CAPACITY { return check(rc); }
        check(unsafe { ntk_batch_submit(self.ctx.0, self.batches[self.cur], &self.params) })?;
        self.in_flight[self.cur] = true;
        self.cur ^= 1;
        if self.in_flight[self.cur] { check(unsafe { ntk_batch_wait(self.ctx.0, self.batches[self.cur]) })?; self.in_flight[self.cur] = false; }
        check(unsafe { ntk_batch_append(self.batches[self.cur], sequence.as_ptr(), sequence.len() as u64, self.params.pre) })
    }
    pub fn finish(mut self) -> Result<Box<NtkResult>, AmdError> {
        check(unsafe { ntk_batch_submit(self.ctx.0, self.batches[self.cur], &self.params) })?;
        self.in_flight[self.cur] = true;
        for i in 0..2 { if self.in_flight[i] { check(unsafe { ntk_batch_wait(self.ctx.0, self.batches[i]) })?; self.in_flight[i] = false; } }
        self.ctx.result()
    }
}
impl Drop for AmdBatchScanner<'_> { fn drop(&mut self) { for b in self.batches { unsafe { ntk_batch_release(self.ctx.0, b) } } } }

/// Record batches shard across the GPUs of a node; the only exchange is ONE ncclAllReduce(ncclUint64, ncclSum) of the
/// accumulators.  One process driving `n` devices:
pub struct AmdComm(*mut NtkComm);
impl AmdComm {
    pub fn all_local(ctxs: &[&AmdContext]) -> Result<Self, AmdError> {
        let raw: Vec<*mut NtkCtx> = ctxs.iter().map(|c| c.0).collect();
        let mut p = std::ptr::null_mut();
        check(unsafe { ntk_comm_init_all(raw.as_ptr(), raw.len() as c_int, &mut p) })?;
        Ok(AmdComm(p))
    }
    /// one process per GPU: rank 0 calls `unique_id`, ships the 128 bytes (MPI, a file, a socket), everyone calls `for_rank`
    pub fn unique_id() -> Result<[u8; NTK_COMM_ID_BYTES], AmdError> {
        let mut id = [0u8; NTK_COMM_ID_BYTES];
        check(unsafe { ntk_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    pub fn for_rank(ctx: &AmdContext, n_ranks: i32, rank: i32, id: &[u8; NTK_COMM_ID_BYTES]) -> Result<Self, AmdError> {
        let mut p = std::ptr::null_mut();
        check(unsafe { ntk_comm_init_rank(ctx.0, n_ranks, rank, id.as_ptr(), &mut p) })?;
        Ok(AmdComm(p))
    }
    /// after the last batch of the run: every context's accumulators become the sum over all GPUs
    pub fn allreduce(&self) -> Result<(), AmdError> { check(unsafe { ntk_allreduce_accumulators(self.0) }) }
}
impl Drop for AmdComm { fn drop(&mut self) { unsafe { ntk_comm_destroy(self.0) } } }

// ---------------------------------------------------------------------------------------------------------------------------------------
// The `Sequence`-trait face (reference src/sequence.rs:156-253), as code rather than as prose in INTEGRATION.md (missing 4).
// The reference's trait hands out iterators per record (`seq.canonical_kmers(k, &rc)`, `seq.bit_kmers(k, canonical)`); a device call per record
// would be launch-latency-bound, so the unit here is the reader's BATCH: `AmdRecords::scan` uploads every record of one `FastxReader` buffer
// once (two bit planes per window start come back, + the dense values for the bit path) and `AmdRecords::record(i)` is a value with the trait's
// method names whose iterators yield exactly the reference's items.  In the crate this file is `mod amd` under `cfg(feature = "amd")`, and
// `impl<'a> Sequence<'a> for AmdRecord<'a>` below is what makes user code that is generic over `Sequence` run on it unchanged: `sequence()`
// is the only required method (src/sequence.rs:158-160); `canonical_kmers` / `bit_kmers` are inherent methods that shadow the provided ones
// for direct callers (the provided ones stay correct - they are the CPU iterators).
// NOT compiled in this repository (no rustc in the image): kept in step with the header by tests/test_abi.py, which diffs the extern block.
// ---------------------------------------------------------------------------------------------------------------------------------------

/// Every record of one reader batch, scanned once on the device for (k, bit-path canonical flag).
pub struct AmdRecords<'b> {
    seq: &'b [u8], offsets: &'b [u64], k: u8,
    bytes: Option<AmdCanonicalKmersPlanes>,   // Sequence::canonical_kmers (byte path; raw-byte strand compare, ties -> rc: src/kmer.rs:121-128)
    bits: Option<AmdBitKmersPlanes>,          // Sequence::bit_kmers (2-bit path; ties stay forward: src/bitkmer.rs:136-143)
}
impl<'b> AmdRecords<'b> {
    /// `seq` + `offsets` (n + 1 entries): record i = seq[offsets[i]..offsets[i + 1]], e.g. the reader's own buffer (no copy is made).
    /// `byte_path` / `bit_path`: which of the two k-mer methods the caller is going to use (each costs one device call for the batch).
    pub fn scan(ctx: &AmdContext, seq: &'b [u8], offsets: &'b [u64], k: u8, byte_path: bool, bit_path: Option<bool>) -> Result<Self, AmdError> {
        let bytes = if byte_path { Some(AmdCanonicalKmersPlanes::new(ctx, seq, offsets, k)?) } else { None };
        let bits = match bit_path { Some(canonical) => Some(AmdBitKmersPlanes::new(ctx, seq, offsets, k, canonical)?), None => None };
        Ok(Self { seq, offsets, k, bytes, bits })
    }
    pub fn len(&self) -> usize { self.offsets.len().saturating_sub(1) }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
    pub fn record(&'b self, i: usize) -> AmdRecord<'b> {
        AmdRecord { batch: self, index: i, seq: &self.seq[self.offsets[i] as usize..self.offsets[i + 1] as usize] }
    }
}

/// One record of the batch: the value user code calls the `Sequence` methods on.
#[derive(Clone, Copy)]
pub struct AmdRecord<'a> { batch: &'a AmdRecords<'a>, index: usize, seq: &'a [u8] }
impl<'a> AmdRecord<'a> {
    /// `Sequence::sequence` (src/sequence.rs:158-160)
    pub fn sequence(&self) -> &'a [u8] { self.seq }
    /// `Sequence::canonical_kmers(k, &rc)` (src/sequence.rs:237-239): the items of `CanonicalKmers` (src/kmer.rs:114-129), from the batch's
    /// bit planes.  `reverse_complement` is the caller's buffer, as in the reference (the slices of reverse-strand items point into it).
    /// Panics like the reference on k = 0 / k > len (src/kmer.rs:91) - here: when the batch was scanned for another k or without the byte path.
    pub fn canonical_kmers(&self, k: u8, reverse_complement: &'a [u8]) -> impl Iterator<Item = (usize, &'a [u8], bool)> + 'a {
        assert!(k == self.batch.k, "the batch was scanned for k = {}", self.batch.k);
        self.batch.bytes.as_ref().expect("AmdRecords::scan(.., byte_path = true, ..)").iter(self.index, self.seq, reverse_complement)
    }
    /// `Sequence::bit_kmers(k, canonical)` (src/sequence.rs:250-252): the items of `BitNuclKmer` (src/bitkmer.rs:97-108).
    pub fn bit_kmers(&self, k: u8, _canonical: bool) -> impl Iterator<Item = (usize, (u64, u8), bool)> + 'a {
        assert!(k == self.batch.k, "the batch was scanned for k = {}", self.batch.k);
        self.batch.bits.as_ref().expect("AmdRecords::scan(.., bit_path = Some(canonical))").iter(self.index)
    }
}
/// In the crate (`src/amd.rs` next to `src/sequence.rs`), feature-gated; `normalize`, `strip_returns`, `reverse_complement`, `kmers` are the
/// trait's provided methods - per-record byte maps that stay on the CPU (on the device they are fused into the scan).
#[cfg(feature = "amd")]
impl<'a> crate::sequence::Sequence<'a> for AmdRecord<'a> {
    fn sequence(&'a self) -> &'a [u8] { self.seq }
}
