"""needletail_amd — MI355X-native k-mer extraction engine behind needletail's Sequence-trait surface.

Only the hot path lives here: csrc/ (HIP kernels + the C ABI of include/needletail_amd.h) and the
host-side mirror of the reference interface.  Importing the package does not need a GPU; calling it does.
"""
from ._lib import (PATH_BITS, PATH_BITS_CANONICAL, PATH_BYTES_CANONICAL, PRE_NONE, PRE_NORMALIZE,
                   PRE_NORMALIZE_IUPAC, PRE_STRIP_RETURNS, NtkError)
from .engine import Batch, Context, default_context
from .parser import (FastxReader, NeedletailError, Record, decode_phred, parse_fastx_file, parse_fastx_stdin, parse_fastx_string, scan_file,
                     write_fasta, write_fastq,
                     scan_file_parallel)
from .sequence import (bit_kmers, bit_kmers_arrays, bit_kmers_batch, canonical_kmers, canonical_kmers_arrays,
                       canonical_kmers_batch, canonical_kmers_planes, CanonicalKmersPlanes, bit_kmers_planes, BitKmersPlanes, kmers, normalize,
                       normalize_opt, normalize_seq, reverse_complement, strip_returns, minimizer, minimizer_batch, canonical, mask_header_tabs,
                       mask_header_utf8, bit_minimizers, quality_mask, bit_reverse_complement, bit_canonical,
                       bitmer_to_bytes, bytes_to_bitmer)

__all__ = [
    "Context", "Batch", "default_context", "NtkError",
    "PATH_BYTES_CANONICAL", "PATH_BITS", "PATH_BITS_CANONICAL",
    "PRE_NONE", "PRE_STRIP_RETURNS", "PRE_NORMALIZE", "PRE_NORMALIZE_IUPAC",
    "parse_fastx_file", "parse_fastx_stdin", "decode_phred", "write_fasta", "write_fastq", "parse_fastx_string", "FastxReader", "Record", "NeedletailError", "scan_file", "scan_file_parallel",
    "normalize", "normalize_opt", "normalize_seq", "strip_returns", "reverse_complement",
    "minimizer", "minimizer_batch", "canonical", "mask_header_tabs", "mask_header_utf8", "bit_minimizers", "quality_mask", "bit_reverse_complement", "bit_canonical", "bitmer_to_bytes",
    "bytes_to_bitmer",
    "kmers", "canonical_kmers", "canonical_kmers_arrays", "bit_kmers", "bit_kmers_arrays", "bit_kmers_batch", "canonical_kmers_batch", "canonical_kmers_planes", "CanonicalKmersPlanes", "bit_kmers_planes", "BitKmersPlanes",
]
