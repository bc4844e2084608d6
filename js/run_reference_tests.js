// Runs the REFERENCE's own mocha test files (cscott/compressjs test/*.js, unchanged) against this drop-in.
//
//   node js/run_reference_tests.js <dir with a test/ subdirectory holding the reference's test files and fixtures>
//
// The test files are not part of this repository: __graft_entry__.build() stages them from /root/reference/test
// into oracle/_ref/reftests/test (git-ignored).  What this file supplies is (1) a ~40-line stand-in for mocha's
// describe/it, (2) a module-resolution hook: require('../') of a test file is js/index.js, require('../lib/Util')
// the three buffer helpers the tests use, require('../lib/HuffmanAllocator') the drop-in's entry.  In test/file.js
// only the codecs of the accelerated path (bzip2, bwtc) run; the other twelve belong to the reference package.
// Prints one JSON line: {files, passed, failed, skipped, failures: [...]}.
'use strict';
var path = require('path'), fs = require('fs'), Module = require('module');
var root = path.resolve(process.argv[2] || '.');
var dropin = require('./index.js');
var utilShim = {                                           // lib/Util.js:248-281, only what test/*.js touches
  makeU8Buffer: function(n) { return new Uint8Array(n); },
  makeS32Buffer: function(n) { return new Int32Array(n); },
  arraycopy: function(dst, src) { for (var i = 0; i < src.length; i++) dst[i] = src[i]; return dst; }
};
var origLoad = Module._load;
Module._load = function(request, parent) {
  if (parent && parent.filename && parent.filename.indexOf(path.join(root, 'test') + path.sep) === 0) {
    if (request === '../' || request === '..') return dropin;
    if (request === '../lib/Util') return utilShim;
    if (request === '../lib/HuffmanAllocator') return dropin.HuffmanAllocator;
  }
  return origLoad.apply(this, arguments);
};

var FILES = process.env.REFTEST_FILES ? process.env.REFTEST_FILES.split(',') :
    ['bwtest.js', 'suftest.js', 'huffman.js', 'bzip2-basic.js', 'bzip2-block.js', 'bzip2-table.js', 'file.js'];
var ONLY_IN_FILE_JS = /^(bzip2|bwtc) /;                     // top-level describes of test/file.js that are on the accelerated path
var res = { files: 0, passed: 0, failed: 0, skipped: 0, failures: [] };
var stack = [], current = null, skipDepth = 0;
global.describe = function(name, fn) {
  var skip = skipDepth > 0 || (current === 'file.js' && stack.length === 0 && !ONLY_IN_FILE_JS.test(name));
  stack.push(name);
  if (skip) skipDepth++;
  try { fn.call({ timeout: function() {} }); } finally { stack.pop(); if (skip) skipDepth--; }
};
global.it = function(name, fn) {
  var title = current + ': ' + stack.concat([name]).join(' / ');
  if (skipDepth > 0) { res.skipped++; return; }
  if (fn.length > 0) { res.failed++; res.failures.push(title + ': asynchronous tests are not supported by this stand-in'); return; }
  try { fn.call({ timeout: function() {} }); res.passed++; }
  catch (e) { res.failed++; res.failures.push(title + ': ' + (e && e.message ? e.message : String(e)).slice(0, 300)); }
};
process.chdir(root);                                        // the tests read 'test/<fixture>'
FILES.forEach(function(f) {
  var p = path.join(root, 'test', f);
  if (!fs.existsSync(p)) { res.failures.push(f + ': not staged'); res.failed++; return; }
  current = f; res.files++;
  try { require(p); } catch (e) { res.failed++; res.failures.push(f + ': ' + (e && e.stack ? e.stack : String(e)).slice(0, 400)); }
});
console.log(JSON.stringify(res));
process.exit(res.failed ? 1 : 0);
