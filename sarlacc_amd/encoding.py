"""Quality encodings: the named numeric vector the reference's R code builds with
.create_encoding_vector (/root/reference/R/qualityMask.R:19-28) and hands to every
quality-aware `.Call` routine."""
import numpy as np


class Encoding:
    """names: bytes of consecutive ASCII characters; errors: matching error probabilities."""

    __slots__ = ("errors", "names")

    def __init__(self, errors, names):
        self.errors = np.ascontiguousarray(errors, dtype=np.float64)
        self.names = names.encode() if isinstance(names, str) else bytes(names)
        if len(self.names) != self.errors.size:
            raise ValueError("encoding vector must be non-empty and named")

    def __len__(self):
        return self.errors.size

    def to_error(self, qual_bytes):
        """vectorised quality_encoding::to_error (src/quality_encoding.cpp:39-47)"""
        q = np.frombuffer(qual_bytes, dtype=np.uint8).astype(np.int64) if isinstance(qual_bytes, (bytes, bytearray)) \
            else np.asarray(qual_bytes, dtype=np.int64)
        idx = q - self.names[0]
        if (idx < 0).any():
            raise ValueError("quality cannot be lower than smallest encoded value")
        return self.errors[np.minimum(idx, self.errors.size - 1)]


def phred_encoding():
    """PhredQuality: '!'..'~' -> 10^(-q/10), q = 0..93 (Biostrings' encoding(PhredQuality()))."""
    q = np.arange(94, dtype=np.float64)
    return Encoding(np.power(10.0, -q / 10.0), bytes(range(33, 127)))


def illumina_encoding():
    """IlluminaQuality: offset 64, q = 0..62 ('@'..'~'), 10^(-q/10) (Biostrings' IlluminaQuality; like the
    Phred table the printable range only: Biostrings' maximum of 99 lies beyond '~')."""
    q = np.arange(63, dtype=np.float64)
    return Encoding(np.power(10.0, -q / 10.0), bytes(range(64, 127)))


def solexa_encoding():
    """SolexaQuality: offset 64, q = -5..62 (';'..'~'); the Solexa score is -10 log10(p / (1 - p)), so
    p = 1 - 1 / (1 + 10^(-q/10)) (as.numeric of a SolexaQuality in Biostrings)."""
    q = np.arange(-5, 63, dtype=np.float64)
    return Encoding(1.0 - 1.0 / (1.0 + np.power(10.0, -q / 10.0)), bytes(range(59, 127)))


QUAL_TYPES = ("phred", "solexa", "illumina")


def encoding_for_qual_type(qual_type="phred"):
    """match.arg(qual.type) + .qual2class + .create_encoding_vector (R/adaptorAlign.R:8,:18-19,:97-99,
    R/qualityMask.R:19-28): the encoding vector of the quality class a FASTQ file is read with.
    Unique prefixes are accepted as match.arg does; a tuple selects its first element (the default)."""
    if isinstance(qual_type, (tuple, list)):
        qual_type = qual_type[0]
    hits = [t for t in QUAL_TYPES if isinstance(qual_type, str) and qual_type and t.startswith(qual_type)]
    if len(hits) != 1:
        raise ValueError("'arg' should be one of 'phred', 'solexa', 'illumina'")
    return hits[0], {"phred": phred_encoding, "solexa": solexa_encoding, "illumina": illumina_encoding}[hits[0]]()


def as_encoding(enc):
    if isinstance(enc, Encoding):
        return enc
    errors, names = enc
    return Encoding(errors, names)


def error_to_phred_char(err):
    """Counterpart of PhredQuality(numeric): error probability -> Phred+33 character(s),
    as used by mockReads (/root/reference/R/mockReads.R:82).  Biostrings rounds
    -10*log10(p) and caps at 99; we cap at 93 ('~') so the result stays printable,
    matching the encoding table above."""
    err = np.asarray(err, dtype=np.float64)
    with np.errstate(divide="ignore"):
        q = np.where(err > 0, np.round(-10.0 * np.log10(np.maximum(err, 1e-300))), 93.0)
    q = np.clip(q, 0, 93).astype(np.uint8)
    return (q + 33).astype(np.uint8)
