"""MI355X-native LZFSE block codec behind lzfse_rust's slice API (encode_bytes / decode_bytes)."""
from .codec import (Context, LzfseDecoder, LzfseEncoder, LzfseError, LzfseReader, LzfseRingDecoder, LzfseRingEncoder, LzfseWriter, LzfseWriterBytes, decode_bytes, decode_size,  # noqa: F401
                    encode_bound, encode_bytes, encode_chunked, decode_chunked, encode_small, device_count)
