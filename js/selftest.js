// node js/selftest.js <golden.json> : compresses a few deterministic inputs through the drop-in
// module and prints sha256 digests as JSON (compared by tests/test_gpu_js.py with the golden file).
'use strict';
var crypto = require('crypto');
var cjs = require('./index.js');
function sha(b) { return crypto.createHash('sha256').update(Buffer.from(b)).digest('hex'); }
var res = {};
res.a1000 = sha(cjs.Bzip2.compressFile(Buffer.alloc(1000, 'a'), null, 9));
res.empty = sha(cjs.Bzip2.compressFile(Buffer.alloc(0)));
var all = Buffer.alloc(256 * 40); for (var i = 0; i < all.length; i++) all[i] = i & 255;
res.bytes40 = sha(cjs.Bzip2.compressFile(all, null, 9));
// LCG(250000, 7) of SURVEY.md 8c
var n = 250000, s = 7, lcg = Buffer.alloc(n);
for (var k = 0; k < n; k++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; lcg[k] = 32 + ((s >>> 16) % 95); }
res.lcg250000 = sha(cjs.Bzip2.compressFile(lcg, null, 1));
var U = Buffer.alloc(7);
res.bwt = [cjs.BWT.bwtransform2(Buffer.from('bcababa'), U, 7, 256), U.toString('ascii')];
var sink = { bytes: [], writeByte: function(b) { this.bytes.push(b); } };
cjs.Bzip2.compressFile({ data: Buffer.from('hello hello hello'), i: 0, readByte: function() { return this.i < this.data.length ? this.data[this.i++] : -1; } }, sink, 9);
res.stream_len = sink.bytes.length;
res.bwtc_a1000 = sha(cjs.BWTC.compressFile(Buffer.alloc(1000, 'a'), null, 9));
res.bwtc_bytes40 = sha(cjs.BWTC.compressFile(all, null, 6));
var U2 = Buffer.alloc(6); res.bwt_linear = [cjs.BWT.bwtransform(Buffer.from('banana'), U2, null, 6, 256), U2.toString('ascii')];
var SA = new Int32Array(6); cjs.BWT.suffixsort(Buffer.from('banana'), SA, 6, 256); res.sa = Array.from(SA);
var U3 = Buffer.alloc(6); cjs.BWT.unbwtransform(Buffer.from('annbaa'), U3, null, 6, 4); res.unbwt = U3.toString('ascii');
var hl = [1, 1, 1, 1, 1]; cjs.HuffmanAllocator.allocateHuffmanCodeLengths(hl, 32); res.huff = hl;
var z = cjs.Bzip2.compressFile(lcg, null, 1);
res.roundtrip = sha(cjs.Bzip2.decompressFile(z)) === sha(lcg);
var tb = []; cjs.Bzip2.table(z, function(p, n) { tb.push([p, n]); }); res.table = tb;
res.block1 = sha(cjs.Bzip2.decompressBlock(z, tb[1][0])) === sha(lcg.slice(tb[0][1], tb[0][1] + tb[1][1]));
var zb = Buffer.from(z); zb[zb.length - 3] ^= 1;
try { cjs.Bzip2.decompressFile(zb); res.badcrc = 'no throw'; } catch (e) { res.badcrc = [e.constructor.name, e.errorCode, e.message.replace(/\(.*\)/, '()')]; }
try { cjs.Bzip2.decompressFile(Buffer.from('BZx9')); res.badmagic = 'no throw'; } catch (e) { res.badmagic = [e.errorCode, e.message]; }
res.sized = cjs.Bzip2.decompressFile(cjs.Bzip2.compressFile(Buffer.from('hello hello')), 11).length;
res.bwtc_roundtrip = sha(cjs.BWTC.decompressFile(cjs.BWTC.compressFile(lcg, null, 7))) === sha(lcg);
try { cjs.Bzip2.compressFile(Buffer.from('x'), null, 0); res.badlevel = 'no throw'; } catch (e) { res.badlevel = e.message; }
// several contexts (here: the same GPU three times) behind the one entry point: same bytes (cjs_bz2_compress_multi)
res.devices = cjs.deviceCount();
var big = Buffer.alloc(3000000); for (var q = 0; q < big.length; q++) big[q] = lcg[(q * 7) % n] ^ (q >> 13 & 15);
var one = sha(cjs.Bzip2.compressFile(big, null, 1));
cjs.configure({ devices: [0, 0, 0], blocksInFlight: 8 });
res.multi_same = sha(cjs.Bzip2.compressFile(big, null, 1)) === one;
cjs.configure({ devices: [0], blocksInFlight: 128 });
console.log(JSON.stringify(res));
