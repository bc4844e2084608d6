// node js/timeprobe.js <raw byte file> [level] [d] : wall time of Bzip2.compressFile (and, with `d`, of Bzip2.decompressFile of its result) through the N-API drop-in (not a test).
'use strict';
var fs = require('fs'), crypto = require('crypto');
var cjs = require('./index.js');
var buf = fs.readFileSync(process.argv[2]);
var level = +(process.argv[3] || 9);
var out = cjs.Bzip2.compressFile(buf, null, level);                // warm-up (contexts, staging buffers)
var best = 1e9;
for (var i = 0; i < 4; i++) {
  var t0 = process.hrtime.bigint();
  out = cjs.Bzip2.compressFile(buf, null, level);
  var dt = Number(process.hrtime.bigint() - t0) / 1e6;
  if (dt < best) best = dt;
}
console.log(JSON.stringify({ bytes: buf.length, out: out.length, ms: +best.toFixed(2), mb_per_s: +(buf.length / best / 1e3).toFixed(1),
                             sha256: crypto.createHash('sha256').update(Buffer.from(out)).digest('hex') }));
// the same calls one per turn of the event loop, as a server makes them: results of earlier turns have been collected, their blocks are reused
if (process.argv.indexOf('loop') > 0) {
  var n = 0, bl = 1e9;
  var turn = function() {
    if (global.gc) global.gc();
    var t2 = process.hrtime.bigint();
    var o2 = cjs.Bzip2.compressFile(buf, null, level);
    var d3 = Number(process.hrtime.bigint() - t2) / 1e6;
    if (d3 < bl) bl = d3;
    if (++n < 8) setImmediate(turn);
    else console.log(JSON.stringify({ per_turn_ms: +bl.toFixed(2), mb_per_s: +(buf.length / bl / 1e3).toFixed(1), out: o2.length }));
  };
  setImmediate(turn);
}
if (process.argv[4] === 'd') {
  var comp = Buffer.from(out.buffer, out.byteOffset, out.length), back = cjs.Bzip2.decompressFile(comp), bd = 1e9;
  for (var j = 0; j < 3; j++) {
    var t1 = process.hrtime.bigint();
    back = cjs.Bzip2.decompressFile(comp);
    var d2 = Number(process.hrtime.bigint() - t1) / 1e6;
    if (d2 < bd) bd = d2;
  }
  console.log(JSON.stringify({ decompress_ms: +bd.toFixed(2), mb_per_s: +(back.length / bd / 1e3).toFixed(1), equal: Buffer.compare(Buffer.from(back.buffer, back.byteOffset, back.length), buf) === 0 }));
}
