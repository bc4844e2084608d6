// Runs the *reference* (cscott/compressjs, mounted read-only at /root/reference) under node to
// produce golden vectors.  Used only by make_golden.py in the build container; never at test time.
// usage: node ref_runner.js jobs.json results.json
'use strict';
var fs = require('fs'), path = require('path'), os = require('os'), crypto = require('crypto');
var REF = process.env.COMPRESSJS_REF || '/root/reference';
// 10-line stand-in for the 'amdefine' dependency (package.json:24), which is not installed here.
var shim = fs.mkdtempSync(path.join(os.tmpdir(), 'amdshim-'));
fs.mkdirSync(path.join(shim, 'node_modules', 'amdefine'), { recursive: true });
fs.writeFileSync(path.join(shim, 'node_modules', 'amdefine', 'index.js'),
  "module.exports=function(m){return function(d,f){if(typeof d==='function'){f=d;d=[];}" +
  "var r=f.apply(null,d.map(function(x){return m.require(x);}));if(r!==undefined)m.exports=r;};};");
require('module').globalPaths.push(path.join(shim, 'node_modules'));
process.env.NODE_PATH = path.join(shim, 'node_modules');
require('module').Module._initPaths();
var cjs = require(path.join(REF, 'main.js'));
var HA = require(path.join(REF, 'lib', 'HuffmanAllocator.js'));
function sha(b) { return crypto.createHash('sha256').update(Buffer.from(b)).digest('hex'); }

var jobs = JSON.parse(fs.readFileSync(process.argv[2]));
var out = [];
jobs.forEach(function(j) {
  var r = { id: j.id, kind: j.kind };
  if (j.kind === 'bz2' || j.kind === 'bwtc') {
    var inp = fs.readFileSync(j.input);
    var codec = j.kind === 'bz2' ? cjs.Bzip2 : cjs.BWTC;
    var t0 = process.hrtime.bigint();
    var o = Buffer.from(codec.compressFile(inp, null, j.level));
    r.seconds = Number(process.hrtime.bigint() - t0) / 1e9;
    r.in_len = inp.length; r.in_sha256 = sha(inp);
    r.out_len = o.length; r.out_sha256 = sha(o); r.level = j.level;
    if (o.length <= 4096 || j.keep) r.out_hex = o.toString('hex');
    if (j.kind === 'bz2') {
      var blocks = [];
      cjs.Bzip2.table(o, function(pos, size) { blocks.push([pos, size]); });
      r.blocks = blocks;
    }
    var back = Buffer.from(codec.decompressFile(o));
    if (Buffer.compare(back, inp) !== 0) throw new Error('reference round trip failed for ' + j.id);
  } else if (j.kind === 'bwt2' || j.kind === 'bwt') {
    var T = fs.readFileSync(j.input), n = T.length;
    var U = Buffer.alloc(n);
    var pidx = j.kind === 'bwt2' ? cjs.BWT.bwtransform2(T, U, n, 256)
                                 : cjs.BWT.bwtransform(T, U, new Int32Array(n), n, 256);
    r.n = n; r.pidx = pidx; r.u_sha256 = sha(U); r.in_sha256 = sha(T);
    if (n <= 256) { r.u_hex = U.toString('hex'); r.in_hex = T.toString('hex'); }
  } else if (j.kind === 'unbwt') {
    // BWT.unbwtransform on an ARBITRARY (T, pidx) pair (not necessarily a BWT): lib/BWT.js:352-363
    var Tu = fs.readFileSync(j.input), nu = Tu.length, Uu = Buffer.alloc(nu);
    cjs.BWT.unbwtransform(Tu, Uu, new Int32Array(nu), nu, j.pidx);
    r.n = nu; r.pidx = j.pidx; r.u_sha256 = sha(Uu); r.in_sha256 = sha(Tu);
    if (nu <= 256) r.u_hex = Uu.toString('hex');
  } else if (j.kind === 'sa') {
    var T2 = fs.readFileSync(j.input), SA = new Int32Array(T2.length);
    cjs.BWT.suffixsort(T2, SA, T2.length, 256);
    r.n = T2.length; r.sa_sha256 = sha(Buffer.from(SA.buffer)); r.in_sha256 = sha(T2);
  } else if (j.kind === 'huff') {
    r.cases = j.cases.map(function(c) {
      var a = c.freq.slice();
      HA.allocateHuffmanCodeLengths(a, c.max_len);
      return { freq: c.freq, max_len: c.max_len, lengths: a };
    });
  } else if (j.kind === 'unbz2' || j.kind === 'unbz2block') {
    // Bzip2.decompressFile / decompressBlock / table of the reference on a (possibly malformed) stream
    var st = fs.readFileSync(j.input);
    r.stream_len = st.length; r.stream_sha256 = sha(st);
    try {
      var dec = j.kind === 'unbz2' ? cjs.Bzip2.decompressFile(st, undefined, !!j.multistream)
                                   : cjs.Bzip2.decompressBlock(st, j.bitpos);
      dec = Buffer.from(dec);
      r.ok = true; r.out_len = dec.length; r.out_sha256 = sha(dec);
      if (j.kind === 'unbz2') {
        var tb = [];
        try { cjs.Bzip2.table(st, function(pos, size) { tb.push([pos, size]); }, !!j.multistream); r.table = tb; }
        catch (e2) { r.table_error = String(e2.message); }
      }
    } catch (e) {
      r.ok = false; r.error_code = (typeof e.errorCode === 'number') ? e.errorCode : null;
      r.message = String(e.message); r.error_type = e.constructor && e.constructor.name;
    }
  } else if (j.kind === 'crc') {
    var CRC32 = require(path.join(REF, 'lib', 'CRC32.js'));
    var d = fs.readFileSync(j.input), c = new CRC32();
    for (var i = 0; i < d.length; i++) c.updateCRC(d[i]);
    r.crc = c.getCRC(); r.in_sha256 = sha(d);
  }
  out.push(r);
});
fs.writeFileSync(process.argv[3], JSON.stringify(out));
