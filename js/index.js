// Drop-in module for the bzip2 path of cscott/compressjs: same export shape as the reference's
// main.js:4-28 for the functions on the accelerated path.
//
//   Bzip2.compressFile(input, [output], [level])   -> MI355X (N-API addon -> C ABI -> HIP kernels)
//   BWT.bwtransform2(T, U, n, [alphabetSize])       -> MI355X
//   BWT.bwtransform / suffixsort / unbwtransform, BWTC.compressFile (levels 6-9) -> MI355X
//   Bzip2.decompressFile / decompressBlock / table   -> MI355X (GPU decoder K7-K9)
//   BWTC.decompressFile                               -> host range decoder + MI355X inverse BWT
//   everything else (the other 12 codecs)
//       -> delegated unchanged to an installed reference package (require('compressjs')), when
//          there is one; otherwise those properties throw.
'use strict';
var path = require('path');
var addon = require(path.join(__dirname, '..', 'build', 'compressjs_amd.node'));
var libPath = process.env.COMPRESSJS_AMD_LIB ||
    path.join(__dirname, '..', 'compressjs_amd', 'libcompressjs_amd.so');
var loaded = addon.load(libPath);

var reference = null;
try { reference = require(process.env.COMPRESSJS_REFERENCE || 'compressjs'); } catch (e) { reference = null; }

var EOF = -1; // Stream.EOF (lib/Stream.js)

// Util.coerceInputStream (lib/Util.js:9-29): buffers as they are; stream objects drained via readByte()
function inputBytes(input) {
  if (input && typeof input.readByte === 'function') {
    var chunks = [], b;
    while ((b = input.readByte()) !== EOF) chunks.push(b & 0xFF);
    return Buffer.from(chunks);
  }
  if (Buffer.isBuffer(input)) return input;
  if (input instanceof Uint8Array) return Buffer.from(input.buffer, input.byteOffset, input.length);
  return Buffer.from(input); // plain Array of byte values
}

// Util.coerceOutputStream (lib/Util.js:85-103)
function deliver(bytes, output) {
  if (!output) {
    // a fresh exact-length Uint8Array (lib/Util.js:96-101).  The addon's result Buffer already owns an ArrayBuffer of exactly
    // that length unless it came from Node's small-buffer pool: then, and only then, copy (10^8-byte input: -5 ms per call)
    var view = new Uint8Array(bytes.buffer, bytes.byteOffset, bytes.length);
    return (bytes.byteOffset === 0 && bytes.buffer.byteLength === bytes.length) ? view : view.slice();
  }
  if (typeof output === 'object' && typeof output.writeByte === 'function') {
    for (var i = 0; i < bytes.length; i++) output.writeByte(bytes[i]);
    if (output.flush) output.flush();
    return output;
  }
  var buf = (typeof output === 'number') ? new Uint8Array(output) : output;
  if (buf.length !== bytes.length) throw new TypeError('outputsize does not match decoded input');
  for (var j = 0; j < bytes.length; j++) buf[j] = bytes[j];
  return buf;
}

function need() {
  if (!loaded) throw new Error('compressjs_amd: ' + libPath + ' could not be loaded (' + addon.lastError() +
                               '); build it with __graft_entry__.build(). There is no JavaScript fallback for the accelerated path.');
}

var Bzip2 = Object.create(null);
Bzip2.compressFile = function(inStream, outStream, props) {          // lib/Bzip2.js:879
  var level = 9;
  if (typeof props === 'number') level = props;
  if (level < 1 || level > 9) throw new Error('Invalid block size multiplier');   // :888-890
  need();
  return deliver(addon.compress(inputBytes(inStream), level), outStream);
};
Bzip2.decompressFile = function(input, output, multistream) {       // lib/Bzip2.js:454,931 (Bunzip.decode)
  need();
  return deliver(addon.decompress(inputBytes(input), !!multistream), output);
};
Bzip2.decompressBlock = function(input, bitPos, output) {            // lib/Bzip2.js:482,932
  need();
  return deliver(addon.decompressBlock(inputBytes(input), bitPos), output);
};
Bzip2.table = function(input, callback, multistream) {               // lib/Bzip2.js:508,933
  need();
  var t = addon.table(inputBytes(input), !!multistream);
  for (var i = 0; i < t.length; i += 2) callback(t[i], t[i + 1]);
};

var BWT = Object.create(null);
// The BWT.* entry points sort ONE block per call and take n <= 2^22 - 1 (the bzip2 path never needs more than
// 900 000, the reference's tests 2 130 640; ranks are packed into 22 bits in the refinement kernels).  The reference has no such limit, so a
// larger n goes to the reference package when it is installed next to this one, else it is a RangeError that
// says so (the C ABI reports CJS_E_ARG, -22).
var BWT_MAX_N = (1 << 22) - 1;
function tooBig(name, n, args) {
  if (n <= BWT_MAX_N) return false;
  if (reference) return true;
  throw new RangeError('BWT.' + name + ': n = ' + n + ' exceeds the ' + BWT_MAX_N + '-byte block limit of the MI355X path (install the reference compressjs next to this package for larger inputs)');
}
BWT.bwtransform2 = function(T, U, n, alphabetSize) {                  // lib/BWT.js:372
  if (tooBig('bwtransform2', n)) return reference.BWT.bwtransform2(T, U, n, alphabetSize);
  if (alphabetSize && alphabetSize > 256) {
    if (reference) return reference.BWT.bwtransform2(T, U, n, alphabetSize);
    throw new Error('only byte alphabets are accelerated');
  }
  need();
  var t = inputBytes(T), u = Buffer.alloc(Math.max(n, 1));
  var pidx = addon.bwtransform2(t, u, n);
  for (var i = 0; i < n; i++) U[i] = u[i];
  return pidx;
};
BWT.bwtransform = function(T, U, A, n, alphabetSize) {               // lib/BWT.js:328
  if (tooBig('bwtransform', n)) return reference.BWT.bwtransform(T, U, A, n, alphabetSize);
  if (alphabetSize && alphabetSize > 256) {
    if (reference) return reference.BWT.bwtransform(T, U, A, n, alphabetSize);
    throw new Error('only byte alphabets are accelerated');
  }
  need();
  var t = inputBytes(T), u = Buffer.alloc(Math.max(n, 1));
  var pidx = addon.bwtransform(t, u, n);
  for (var i = 0; i < n; i++) U[i] = u[i];
  return pidx;
};
BWT.suffixsort = function(T, SA, n, alphabetSize) {                  // lib/BWT.js:305
  if (tooBig('suffixsort', n)) return reference.BWT.suffixsort(T, SA, n, alphabetSize);
  if (alphabetSize && alphabetSize > 256) {
    if (reference) return reference.BWT.suffixsort(T, SA, n, alphabetSize);
    throw new Error('only byte alphabets are accelerated');
  }
  need();
  var sa = (SA instanceof Int32Array) ? SA : new Int32Array(Math.max(n, 1));
  addon.suffixsort(inputBytes(T), sa, n);
  if (sa !== SA) for (var i = 0; i < n; i++) SA[i] = sa[i];
  return 0;
};
BWT.unbwtransform = function(T, U, LF, n, pidx) {                   // lib/BWT.js:352 (LF: scratch, unused here)
  if (tooBig('unbwtransform', n)) return reference.BWT.unbwtransform(T, U, LF, n, pidx);
  need();
  var t = inputBytes(T), u = Buffer.alloc(Math.max(n, 1));
  addon.unbwtransform(t, u, n, pidx);
  for (var i = 0; i < n; i++) U[i] = u[i];
};

// require('compressjs/lib/HuffmanAllocator') (lib/HuffmanAllocator.js:199-226)
var HuffmanAllocator = Object.create(null);
HuffmanAllocator.allocateHuffmanCodeLengths = function(array, maximumLength) {
  need();
  var a = Float64Array.from(array);
  addon.huffLengths(a, maximumLength);
  for (var i = 0; i < a.length; i++) array[i] = a[i];
};

var BWTC = Object.create(null);
BWTC.MAGIC = 'bwtc';
BWTC.compressFile = function(inStream, outStream, props) {           // lib/BWTC.js:12
  var level = (typeof props === 'number' && props >= 1 && props <= 9) ? props : 9;   // :16-19
  need();
  var known = !(inStream && typeof inStream.readByte === 'function') || ('size' in inStream && inStream.size >= 0);
  var bytes = inputBytes(inStream);
  return deliver(addon.bwtcCompress(bytes, level, known ? bytes.length : -1), outStream);   // lib/Util.js:119-124
};
BWTC.decompressFile = function(inStream, outStream) {               // lib/BWTC.js:141
  need();
  var bytes = inputBytes(inStream);
  return deliver(addon.bwtcDecompress(bytes), outStream);
};

// Not in the reference (which is single-threaded JavaScript): which GPUs Bzip2.compressFile uses and how many bzip2
// blocks are in flight per GPU.  configure({devices: [0, 1, 2, 3], blocksInFlight: 128}); devices: 'all' = every visible
// one.  Also settable without code through COMPRESSJS_AMD_DEVICES ("all" | "0,1,...") and COMPRESSJS_AMD_BLOCKS.  With
// more than one device the input's segments go round-robin to the devices (cjs_bz2_compress_multi, SURVEY.md 8e);
// the bytes returned are the same.  Returns the number of visible devices.
function configure(opts) {
  need();
  opts = opts || {};
  var devs = opts.devices;
  if (devs === 'all') { devs = []; for (var i = 0, n = addon.deviceCount(); i < n; i++) devs.push(i); }
  return addon.configure(Array.isArray(devs) ? devs : null, opts.blocksInFlight | 0);
}

var out = { version: '0.2.0-mi355x', configure: configure, deviceCount: function() { need(); return addon.deviceCount(); }, Bzip2: Object.freeze(Bzip2), BWT: Object.freeze(BWT), BWTC: Object.freeze(BWTC),
            // not in the reference's main.js facade; the reference reaches it by path (test/huffman.js:3)
            HuffmanAllocator: Object.freeze(HuffmanAllocator) };
if (reference) {
  Object.keys(reference).forEach(function(k) { if (!(k in out)) out[k] = reference[k]; });
}
module.exports = Object.freeze(out);
