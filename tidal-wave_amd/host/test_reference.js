// test_reference.js — plain-node restatement of the reference's mocha suite (/root/reference/test/index.coffee:
// 4 cases) against this addon.  Needs a HIP device (the engine has no CPU path).  Fixture tree:
// tests/golden/tree (the reference's PNG and JPEG fixture files, verbatim).
'use strict';
var TidalWave = require('./index');
var assert = require('assert');
var Path = require('path');
var FIX = Path.resolve(__dirname, '../../tests/golden/tree');
var GOLD = require(Path.resolve(__dirname, '../../tests/golden/expected_responses.json'));

function createWithExpectDir(rev) {
  return TidalWave.create(Path.resolve(FIX, rev), { expectDir: Path.resolve(FIX, 'expected') });
}
function expectOK(data, rev, sc, file, h, w) {
  delete data.time;
  assert.deepStrictEqual(data, {
    status: 'OK', span: 10, threshold: 5,
    expect_image: Path.join(FIX, 'expected', sc, file), target_image: Path.join(FIX, rev, sc, file),
    height: h, width: w, vector: [] });
}
var tests = [
  ['should report nothing on "revision1"', function(done) {
    var t = createWithExpectDir('revision1');
    t.on('data', function(data) {
      if (~data.target_image.indexOf('capture1')) expectOK(data, 'revision1', 'scenario1', 'capture1.jpg', 279, 280);
      else if (~data.target_image.indexOf('capture2')) expectOK(data, 'revision1', 'scenario2', 'capture2.png', 117, 180);
      else assert.fail('Cannot be here.');
    });
    t.on('error', function(e) { assert.fail(JSON.stringify(e)); });
    t.on('finish', function(report) { assert.deepStrictEqual(report, { request: 2, data: 2, error: 0 }); done(); });
  }],
  ['should report something on "revision2"', function(done) {
    var t = createWithExpectDir('revision2');
    t.on('data', function(data) {
      if (~data.target_image.indexOf('capture1')) expectOK(data, 'revision2', 'scenario1', 'capture1.jpg', 279, 280);
      else if (~data.target_image.indexOf('capture2')) {
        delete data.time;
        assert.deepStrictEqual(Object.keys(data), ['status', 'span', 'threshold', 'expect_image', 'target_image', 'height', 'width', 'vector']);
        assert.deepStrictEqual(data, {
          status: 'SUSPICIOUS', span: 10, threshold: 5,
          expect_image: Path.join(FIX, 'expected/scenario2/capture2.png'),
          target_image: Path.join(FIX, 'revision2/scenario2/capture2.png'),
          height: 117, width: 180, vector: GOLD.revision2_capture2.vector });   // the 24 golden vectors
      } else assert.fail('Cannot be here.');
    });
    t.on('error', function(e) { assert.fail(JSON.stringify(e)); });
    t.on('finish', function(report) { assert.deepStrictEqual(report, { request: 2, data: 2, error: 0 }); done(); });
  }],
  ['should never report on "__NOT_EXISTS__"', function(done) {
    var t = createWithExpectDir('__NOT_EXISTS__');
    t.on('data', function() { assert.fail('boom.'); });
    t.on('finish', function(report) { assert.deepStrictEqual(report, { request: 0, data: 0, error: 0 }); done(); });
  }],
  ["should start with passing 'getExpectedPath' option", function(done) {
    var counter = 0;
    var t = TidalWave.create(Path.resolve(FIX, 'revision2'), {
      getExpectedPath: function(shortPath) { return Path.resolve(FIX, 'revision1', shortPath); } });
    t.on('data', function() { counter++; });
    t.on('finish', function(report) {
      assert.strictEqual(counter, 2);
      assert.deepStrictEqual(report, { request: 2, data: 2, error: 0 });
      done();
    });
  }]
];
var only = process.argv[2] === 'nogpu' ? [2] : [0, 1, 2, 3];
(function next(i) {
  if (i >= only.length) { console.log('all ' + only.length + ' reference tests passed'); return; }
  var tc = tests[only[i]];
  var timer = setTimeout(function() { console.error('TIMEOUT: ' + tc[0]); process.exit(2); }, 30000);
  tc[1](function() { clearTimeout(timer); console.log('ok - ' + tc[0]); next(i + 1); });
})(0);
